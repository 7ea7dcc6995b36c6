"""CPU twin of tests/test_gpu_dropin.py: the same plain-C NLopt client (tests/dropin/dropin_demo.c) run against the REAL
reference library and against the product's C sources over the emulated device layer (oracle/libnlopt_amd_emu.so).  On that
layer every kernel's job is done in the reference's operation order with the host's libm, so the two printouts must be
identical character for character — result, minimum, argmin, evaluation count, callback count and the hash of every x the
callback saw — for the host-callback path and for the registered-objective path alike.  Exercises the C ABI, the API shell and
the host drivers on a machine without a GPU."""
import os

import pytest

import _oracle as O
from test_gpu_dropin import run, REF, fields

EMU = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "libnlopt_amd_emu.so")
need = pytest.mark.skipif(not (O.have_ref() and os.path.exists(EMU)), reason="oracle/_ref or the emulated library not built")


@need
@pytest.mark.parametrize("alg,n,pop,maxeval,seed", [(19, 12, 150, 3000, 42), (19, 40, 0, 2500, 7), (19, 3, 0, 800, 1),
                                                    (35, 10, 60, 1200, 5), (35, 6, 0, 1500, 11), (42, 8, 30, 1500, 3), (42, 5, 0, 900, 9),
                                                    (20, 4, 0, 1500, 5), (22, 3, 10, 1200, 8), (25, 6, 0, 400, 1)])      # GN_MLSL(_LDS) with LN_COBYLA, LN_COBYLA
def test_same_client_same_output(alg, n, pop, maxeval, seed):
    ref = run(REF, alg, n, pop, maxeval, seed)
    emu = run(EMU, alg, n, pop, maxeval, seed)
    assert ref == emu, "\nreference: %s\nemulated : %s" % (ref, emu)
    assert fields(emu)["callbacks"] == fields(emu)["numevals"]


@need
@pytest.mark.parametrize("alg,n,pop,maxeval,seed", [(19, 64, 2000, 6000, 42), (35, 9, 40, 800, 2), (42, 7, 25, 1000, 4)])
def test_registered_objective_path_reaches_the_same_line(alg, n, pop, maxeval, seed):
    """`device`: the client registers the library's own objective — no callback reaches the client, the rest of the line is the
    reference's"""
    rf = fields(run(REF, alg, n, pop, maxeval, seed))
    ef = fields(run(EMU, alg, n, pop, maxeval, seed, "device"))
    assert ef["callbacks"] == "0"
    for k in rf:
        if k not in ("callbacks", "xhash"):
            assert rf[k] == ef[k], (k, rf[k], ef[k])
