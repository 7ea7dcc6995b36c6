"""-m gpu: ordinary host callbacks (nlopt_func) through LD_LBFGS, LD_MMA and every MLSL variant on the HIP kernels (VERDICT r1
item 3).  The same drawn clients as tests/test_host_callbacks_emulated.py, issued to the REAL reference and to libnlopt_amd.so
in exact-order mode: every callback invocation (x bit for bit, gradient requested or not, in order), result code, minimum,
argmin and evaluation count must be identical — force_stop from inside the callback, x_weights, xtol_abs (incl. all zeros),
maximisation, LD_MMA's parameters and its uncounted gradient call included."""
import ctypes as C

import numpy as np
import pytest

import _oracle as O
import nlopt_amd
import test_host_callbacks_emulated as T

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")]


@pytest.fixture(scope="module")
def libs():
    assert nlopt_amd.device_count() > 0
    return T.bind(O.ref()), T.bind(C.CDLL(nlopt_amd.LIB_PATH))


def play_exact(L, draw, algs):
    """T.play with the generic parameter amd_exact_dot = 1 on every object created (the reference stores and ignores it)"""
    real_create = L.nlopt_create

    class Shim:
        def __getattr__(self, name):
            return getattr(L, name)

        def nlopt_create(self, alg, n):
            o = real_create(alg, n)
            L.nlopt_set_param(o, b"amd_exact_dot", 1.0)
            return o
    return T.play(Shim(), draw, algs)


@pytest.mark.parametrize("draw", range(60))
def test_local_optimisers_with_host_callbacks_on_the_device(libs, draw):
    R, A = libs
    T.same(play_exact(R, draw, [T.LD_LBFGS, T.LD_MMA]), play_exact(A, draw, [T.LD_LBFGS, T.LD_MMA]), draw)


@pytest.mark.parametrize("draw", range(60))
def test_mlsl_with_host_callbacks_on_the_device(libs, draw):
    R, A = libs
    algs = [T.G_MLSL, T.G_MLSL_LDS, T.GD_MLSL, T.GD_MLSL_LDS]
    T.same(play_exact(R, 1000 + draw, algs), play_exact(A, 1000 + draw, algs), 1000 + draw)


def test_lbfgs_host_callback_runs_on_the_device_at_n4096():
    """config 4's dimension with the client's own callback: 320 history pairs on the device, f and gradient on the host"""
    import test_gpu_exact_local as X
    a = X.run_amd(nlopt_amd.LD_LBFGS, "ackley", 4096, True, ftol_rel=1e-8, exact=False)
    p = O.run_port_lbfgs("ackley", 4096, ftol_rel=1e-8)
    assert a["ret"] == p["ret"] and abs(a["nevals"] - p["nevals"]) <= 4
    assert abs(a["minf"] - p["minf"]) <= 1e-8 * abs(p["minf"])
