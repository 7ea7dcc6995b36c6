"""CPU twin of tests/test_gpu_stops.py: nlopt_force_stop() raised from inside the objective, the same client code against the REAL
reference and against the product over the emulated device — FORCED_STOP after exactly as many evaluations, and (everything on
that layer being in the reference's operation order) exactly the reference's best point, for ISRES and ESCH too."""
import ctypes as C
import os

import numpy as np
import pytest

import _oracle as O
from test_gpu_stops import run_lib

EMU = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "libnlopt_amd_emu.so")
pytestmark = pytest.mark.skipif(not (O.have_ref() and os.path.exists(EMU)), reason="oracle/_ref or the emulated library not built")


@pytest.mark.parametrize("alg,pop,stop_after", [(19, 30, 10), (19, 30, 31), (19, 30, 200), (35, 25, 7), (35, 25, 25), (35, 25, 90), (42, 12, 5), (42, 12, 70),
                                                (42, 0, 41), (19, 0, 1)])
def test_force_stop_from_the_callback(alg, pop, stop_after):
    r = run_lib(O.ref(), alg, 4, pop, 3, stop_after)
    a = run_lib(C.CDLL(EMU), alg, 4, pop, 3, stop_after)
    assert a[0] == r[0] == -5
    assert a[1:4] == r[1:4] or (a[1:3] == r[1:3] and np.isinf(a[3]) and np.isinf(r[3]))
    assert np.array_equal(a[4], r[4])
