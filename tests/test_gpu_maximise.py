"""-m gpu: nlopt_set_max_objective with a registered device objective keeps CRS2_LM / ISRES / ESCH on the device (the evaluation
kernels deliver -f: NLA_OBJ_NEGATE) — against the real reference, which minimises the flipped host callback (optimize.c:1014-1024).
CRS2_LM: nothing on the way to x involves libm, bit for bit; ISRES / ESCH: same result code and evaluation count, optimum to rounding."""
import ctypes as C

import numpy as np
import pytest

import _oracle as O
import nlopt_amd
import test_api_differential as T

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")]


@pytest.mark.parametrize("first", [0, 30])
def test_population_algorithms_maximise_on_the_device(first):
    assert nlopt_amd.device_count() > 0
    P = O.port()
    ref = O.ref()
    ref.orc_objective = P.orc_objective
    A = T.bind(C.CDLL(nlopt_amd.LIB_PATH))
    for draw in range(first, first + 30):
        r = T.play_population_max(T.bind(ref), draw, "orc_objective")
        a = T.play_population_max(A, draw, "nlopt_amd_objective")
        obj, n, alg, maximise, fixed = a["cfg"]
        assert a["log"] == r["log"] and a["ret"] == r["ret"], "draw %d %r: ret %d / %d" % (draw, a["cfg"], a["ret"], r["ret"])
        # a maximum on the boundary of the box collects clamped, nearly equal points: device libm (1 ulp from glibc here and there)
        # can then turn an accept into a reject, and CRS2_LM tests maxeval only after a rejection (crs.c:136) — a few evaluations of slack
        assert abs(a["nev"] - r["nev"]) <= 4, "draw %d %r: nev %d / %d" % (draw, a["cfg"], a["nev"], r["nev"])
        assert abs(a["minf"] - r["minf"]) <= 1e-7 * max(abs(r["minf"]), 1.0), (draw, a["cfg"], a["minf"], r["minf"])
        if alg == 19 and a["nev"] == r["nev"]:
            assert np.array_equal(a["x"], r["x"]), (draw, a["cfg"])          # CRS2_LM: nothing on the way to x involves libm


@pytest.mark.parametrize("case", T.MAX_HOST_CASES, ids=lambda c: "alg%d_%s_n%d_hosteval%d" % c)
def test_maximising_where_the_run_calls_f_on_the_host(case):
    """round-2 advisor (high): with a host constraint, amd_host_eval or constrained LD_MMA the run called the unflipped callback
    and MINIMISED.  These runs take the exact host-callback path: identical to the reference."""
    P = O.port()
    ref = O.ref()
    ref.orc_objective = P.orc_objective
    r = T.play_max_with_host_parts(T.bind(ref), case, "orc_objective")
    a = T.play_max_with_host_parts(T.bind(C.CDLL(nlopt_amd.LIB_PATH)), case, "nlopt_amd_objective")
    assert a["log"] == r["log"] and a["ret"] == r["ret"], (case, a["ret"], r["ret"])
    assert a["maxf"] > 0 and a["nev"] == r["nev"]
    if case[0] == 35:        # ISRES: candidates go through exp / log on the device (1 ulp from glibc here and there)
        assert abs(a["maxf"] - r["maxf"]) <= 1e-9 * max(abs(r["maxf"]), 1.0)
    else:
        assert a["maxf"] == r["maxf"] and np.array_equal(a["x"], r["x"])
