"""-m gpu twin of tests/test_cobyla_differential.py: NLOPT_LN_COBYLA and NLOPT_GN_MLSL(_LDS) with its default local optimiser
through libnlopt_amd.so on the MI355X against the real reference — identical objective calls, results and counts (COBYLA runs on the
host; MLSL's samples, distances and bookkeeping on the device)."""
import ctypes as C

import numpy as np
import pytest

import _oracle as O
import nlopt_amd
import test_cobyla_differential as T

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")]


def test_cobyla_on_the_product_library_is_the_references_run():
    assert nlopt_amd.device_count() > 0
    R, A = T.more_bind(O.ref()), T.more_bind(C.CDLL(nlopt_amd.LIB_PATH))
    for draw in range(0, 60):
        T.same(T.play_cobyla(R, draw), T.play_cobyla(A, draw), draw)


def test_gn_mlsl_default_local_optimiser_on_the_device_library_is_the_references_run():
    assert nlopt_amd.device_count() > 0
    R, A = T.more_bind(O.ref()), T.more_bind(C.CDLL(nlopt_amd.LIB_PATH))
    for draw in range(0, 40):
        T.same(T.play_mlsl(R, draw), T.play_mlsl(A, draw), draw)


@pytest.mark.parametrize("alg,obj,n", [(T.GN_MLSL_LDS, "rastrigin", 3), (T.GN_MLSL, "ackley", 2), (T.GN_MLSL_LDS, "griewank", 5)])
def test_gn_mlsl_with_a_registered_device_objective_takes_the_exact_host_path(alg, obj, n):
    """a compiled-in device objective under GN_MLSL: COBYLA is a host algorithm, so the whole run uses the objective's host twin —
    and is the reference's run evaluation by evaluation (same callback pointer given to both libraries)"""
    L = nlopt_amd.lib()
    fptr = nlopt_amd.objective(obj)
    lo, hi = nlopt_amd.objective_box(obj)
    out = []
    for lib in (T.more_bind(O.ref()), T.more_bind(C.CDLL(nlopt_amd.LIB_PATH))):
        opt = lib.nlopt_create(alg, n)
        lb, ub = np.full(n, lo), np.full(n, hi)
        lib.nlopt_set_lower_bounds(opt, T.dp(lb))
        lib.nlopt_set_upper_bounds(opt, T.dp(ub))
        lib.nlopt_set_min_objective(opt, C.cast(fptr, C.c_void_p), None)
        lib.nlopt_set_xtol_rel(opt, 1e-5)
        lib.nlopt_set_maxeval(opt, 700)
        lib.nlopt_srand(77)
        x = np.linspace(0.3 * lo, 0.4 * hi, n)
        minf = C.c_double(0)
        ret = lib.nlopt_optimize(opt, T.dp(x), C.byref(minf))
        out.append((ret, minf.value, x.copy(), lib.nlopt_get_numevals(opt)))
        lib.nlopt_destroy(opt)
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1] and np.array_equal(out[0][2], out[1][2]) and out[0][3] == out[1][3], out
    assert L is not None
