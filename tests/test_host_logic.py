"""Host-side logic of the product, testable without a GPU:
 * the speculate/commit CRS driver (nlopt_amd/csrc/crs_driver.c) run over a CPU emulation of the
   device engine (oracle/port_emu_engine.c) must reproduce the oracle's serial trace exactly, for
   any speculation depth;
 * MT19937 host generator + GF(2) jump-ahead machinery (nlopt_amd/csrc/mt_host.c)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import _oracle as O
import nlopt_amd

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "crs_golden.json")))
TR = ("f", "row", "kind", "accepted")


def same_trace(a, b):
    return len(a) == len(b) and all(np.array_equal(a[k], b[k]) for k in TR)


@pytest.mark.parametrize("name", sorted(GOLD))
def test_driver_over_emulated_engine_matches_golden(name):
    g = GOLD[name]
    kw = dict(g["kwargs"])
    r = O.run_emu_crs(g["obj"], g["n"], g["pop"], g["seed"], trace_cap=200000, **kw)
    assert r["ret"] == g["ret"] and r["nevals"] == g["nevals"]
    assert float(r["minf"]).hex() == g["minf"]
    assert [float(v).hex() for v in r["x"]] == g["x"]
    p = O.run_port_crs(g["obj"], g["n"], g["pop"], g["seed"], trace_cap=200000, **kw)
    assert same_trace(p["trace"], r["trace"])
    assert p["words"] == r["words"]


@pytest.mark.parametrize("max_slots,max_spec", [(1, 0), (3, 0), (0, 1), (0, 2), (0, 37), (5, 1024), (0, 0)])
def test_trace_is_invariant_under_speculation_depth(max_slots, max_spec):
    base = O.run_port_crs("rastrigin", 24, 300, 17, maxeval=4000, trace_cap=10000)
    r = O.run_emu_crs("rastrigin", 24, 300, 17, maxeval=4000, trace_cap=10000, max_slots=max_slots, max_spec=max_spec)
    assert r["ret"] == base["ret"] and r["nevals"] == base["nevals"]
    assert same_trace(base["trace"], r["trace"])
    assert np.array_equal(base["x"], r["x"]) and base["minf"] == r["minf"]
    st = r["stats"]
    assert st["evals_init"] + st["evals_trial"] + st["evals_mutation"] == r["nevals"]
    assert st["slots_launched"] >= st["slots_used"]


@pytest.mark.parametrize("draw", range(40))
def test_random_configurations_over_the_emulated_engine(draw):
    """drawn objective / dimension / population / seed / stopping rule / speculation limits / window factor / host-callback mode:
    the product's driver must replay the oracle's serial trace whatever the window does"""
    rng = np.random.default_rng(77 + draw)
    obj = ["rastrigin", "ackley", "griewank", "rosenbrock", "levy", "sphere"][int(rng.integers(6))]
    n = int(rng.integers(1, 40))
    pop = int(rng.integers(n + 1, 12 * n + 20))
    seed = int(rng.integers(1, 2 ** 31))
    kw = dict(maxeval=int(rng.integers(pop + 20, pop + 3000)))
    r = rng.random()
    if r < 0.25:
        kw["ftol_rel"] = 10.0 ** -int(rng.integers(2, 8))
    elif r < 0.4:
        kw["xtol_rel"] = 10.0 ** -int(rng.integers(2, 6))
    elif r < 0.5:
        kw["ftol_abs"] = 10.0 ** -int(rng.integers(1, 6))
    base = O.run_port_crs(obj, n, pop, seed, trace_cap=20000, **kw)
    emu_kw = dict(max_slots=int(rng.choice([0, 0, 1, 2, 7, 64])), max_spec=int(rng.choice([0, 0, 1, 3, 50])),
                  window_factor=float(rng.choice([0.0, 0.5, 1.0, 1.5, 3.0, 20.0])), host_eval=bool(rng.random() < 0.25))
    e = O.run_emu_crs(obj, n, pop, seed, trace_cap=20000, **kw, **emu_kw)
    assert (e["ret"], e["nevals"], e["words"]) == (base["ret"], base["nevals"], base["words"]), emu_kw
    assert same_trace(base["trace"], e["trace"]), emu_kw
    assert np.array_equal(base["x"], e["x"]) and base["minf"] == e["minf"]


def test_host_callback_path_matches_serial_reference_order():
    base = O.run_port_crs("levy", 6, 80, 3, maxeval=2500, trace_cap=5000)
    r = O.run_emu_crs("levy", 6, 80, 3, maxeval=2500, trace_cap=5000, host_eval=True)
    assert same_trace(base["trace"], r["trace"]) and r["stats"]["slots_launched"] == r["stats"]["slots_used"]


def test_tiny_population_where_worst_list_covers_everything():
    # N = n+1 = 4: the per-round worst list is the whole population, the best row included
    base = O.run_port_crs("sphere", 3, 4, 5, maxeval=600, trace_cap=2000)
    r = O.run_emu_crs("sphere", 3, 4, 5, maxeval=600, trace_cap=2000)
    assert same_trace(base["trace"], r["trace"]) and np.array_equal(base["x"], r["x"])


# ---- MT19937 host side -------------------------------------------------------------------------
def test_product_generator_matches_oracle_generator():
    L, P = nlopt_amd.lib(), O.port()
    L.nlopt_srand(20240922)
    P.orc_srand(20240922)
    for i in range(5000):
        assert L.nla_genrand_int32() == P.orc_genrand_int32()
    for i in range(500):
        assert L.nlopt_urand(-1.0, 3.0) == P.orc_urand(-1.0, 3.0)
        assert L.nlopt_iurand(4097) == P.orc_iurand(4097)
        assert L.nlopt_nrand(0.0, 1.0) == P.orc_nrand(0.0, 1.0)


def test_characteristic_polynomial_has_the_known_weight():
    L = nlopt_amd.lib()
    exps = C.POINTER(C.c_int)()
    nterms = L.nla_mt_charpoly_terms(C.byref(exps))
    e = [exps[i] for i in range(nterms)]
    assert nterms == 135 and e[0] == 0 and e[-1] == 19937        # weight of MT19937's characteristic polynomial


def _regen_ref(mt, times):
    L = nlopt_amd.lib()
    a = mt.copy()
    L.nla_mt_regen.argtypes = [C.c_void_p]
    for _ in range(times):
        L.nla_mt_regen(a.ctypes.data)
    return a


def test_jump_polynomials_equal_plain_regeneration():
    L = nlopt_amd.lib()
    L.nlopt_srand(777)
    mt = np.zeros(624, dtype=np.uint32)
    cons = C.c_int()
    L.nla_mt_export(mt.ctypes.data, C.byref(cons))
    assert cons.value == 0
    out = np.zeros(624, dtype=np.uint32)
    # pow2 table entry k jumps 2^k regenerations
    for k in (0, 1, 5):
        L.nla_mt_apply_jump_host(L.nla_mt_jump_poly_pow2(k), mt.ctypes.data, out.ctypes.data)
        assert np.array_equal(out, _regen_ref(mt, 1 << k))
    # arbitrary word offset, not a multiple of 624: compare word windows
    J = 624 * 3 + 77
    g = np.zeros(312, dtype=np.uint64)
    L.nla_mt_jump_poly_words(J, g.ctypes.data)
    L.nla_mt_apply_jump_host(g.ctypes.data, mt.ctypes.data, out.ctypes.data)
    seq = np.concatenate([_regen_ref(mt, r) for r in range(6)])
    assert np.array_equal(out, seq[J:J + 624])
    # binary decomposition with plain regeneration for the low bits
    L.nla_mt_advance_blocks_host(mt.ctypes.data, 4096 + 3, out.ctypes.data)
    assert np.array_equal(out, _regen_ref(mt, 4099))


def test_sobol_points_by_index_equal_the_stateful_walk():
    """the product computes Sobol point k directly from gray(k) (nlopt_amd/csrc/sobol.c); the oracle walks the
    reference's stateful generator (oracle/port_sobol.c, pinned to the real one): same doubles"""
    import ctypes as C
    import nlopt_amd
    L = nlopt_amd.lib()
    L.nla_sobol_directions.argtypes = [C.c_uint, C.c_void_p]
    L.nla_sobol_point01.argtypes = [C.c_uint, C.c_void_p, C.c_uint32, C.POINTER(C.c_double)]
    L.nla_sobol_skip_count.argtypes = [C.c_uint]
    L.nla_sobol_skip_count.restype = C.c_uint32
    for sdim, count in ((1, 70), (5, 600), (1111, 130)):
        V = np.zeros(32 * sdim, dtype=np.uint32)
        assert L.nla_sobol_directions(sdim, V.ctypes.data) == 1
        want = O.port_sobol_points(sdim, 0, count)
        x = np.zeros(sdim)
        for k in (1, 2, 3, 4, 7, 8, 63, 64, count - 1, count):
            L.nla_sobol_point01(sdim, V.ctypes.data, k, O.dptr(x))
            assert np.array_equal(x, want[k - 1]), (sdim, k)
    # after nlopt_sobol_skip(s, n, .) the next point is number skip_count(n) + 1
    for n_skip in (1, 2, 3, 17, 64, 65, 10 * 30 + 50):
        k = L.nla_sobol_skip_count(n_skip)
        V = np.zeros(32 * 3, dtype=np.uint32)
        L.nla_sobol_directions(3, V.ctypes.data)
        x = np.zeros(3)
        L.nla_sobol_point01(3, V.ctypes.data, k + 1, O.dptr(x))
        assert np.array_equal(x, O.port_sobol_points(3, n_skip, 1)[0])
    assert L.nla_sobol_directions(1112, np.zeros(32 * 1112, dtype=np.uint32).ctypes.data) == 0


def test_host_callbacks_are_bit_identical_to_the_oracles():
    """the objective zoo is one source (objfuncs.h) compiled into the product and into the oracle; glibc's sincos() rounds
    differently from cos() / sin() for some arguments and gcc merges the calls depending on optimisation level, which once made the
    two builds differ in the last bit (f = 100.75937443450825 vs ...823 for Rastrigin at a 4-d point).  Value and gradient, with
    and without a gradient request, must agree exactly."""
    FT = C.CFUNCTYPE(C.c_double, C.c_uint, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)
    L, P = nlopt_amd.lib(), O.port()
    L.nlopt_amd_objective.restype = C.c_void_p
    L.nlopt_amd_objective.argtypes = [C.c_int]
    P.orc_objective.restype = C.c_void_p
    P.orc_objective.argtypes = [C.c_int]
    rng = np.random.default_rng(0)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    x0 = np.array([4.2152877590035569, -3.5179279536925359, 0.59393502861624903, 4.1945617962291282])
    assert FT(L.nlopt_amd_objective(0))(4, dp(x0), None, None) == FT(P.orc_objective(0))(4, dp(x0), None, None) == 100.75937443450823
    for oid in range(6):
        fa, fb = FT(L.nlopt_amd_objective(oid)), FT(P.orc_objective(oid))
        for _ in range(400):
            n = int(rng.integers(2, 40))
            x = rng.uniform(-6, 6, n)
            ga, gb = np.zeros(n), np.zeros(n)
            va, vb = fa(n, dp(x), None, None), fb(n, dp(x), None, None)
            assert va == vb == fa(n, dp(x), dp(ga), None) == fb(n, dp(x), dp(gb), None)
            assert np.array_equal(ga, gb)


def test_chain_resolver_wavefront_takes_the_sequential_decisions():
    """hip/crs_chain_resolver.h (the accept / reject chain of a device-resolved window advanced by one wavefront
    out of registers) compiled by g++ with the wavefront primitives emulated — 64 threads in lockstep, a feeder thread publishing the
    slots' records out of order and partly only after the chain has passed an earlier slot: the rowstate words, next / pk and the final
    counters equal the sequential statement of crs_trial's decisions (crs.c:125-156) for drawn windows with ties, NaNs, values landing among
    the worst rows, new bests and lists shorter than the window; the watchdog halts a window nobody evaluates
    (tools/chain_resolver_check.cpp).  The device's memory model is not what this checks: tests/test_gpu_kernels.py, tests/test_gpu_crs_windows.py."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not shutil.which("g++"):
        pytest.skip("no g++ here")
    out = os.path.join(root, "tools", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "chain_resolver_check.%d" % os.getpid())
    subprocess.run(["g++", "-O1", "-std=c++17", "-pthread", "-I", os.path.join(root, "nlopt_amd", "csrc", "hip"),
                    os.path.join(root, "tools", "chain_resolver_check.cpp"), "-o", exe], check=True)
    try:
        for seed in (1, 2):
            r = subprocess.run([exe, "120", str(seed)], capture_output=True, text=True, timeout=600)
            assert r.returncode == 0 and r.stdout.startswith("ok 120"), r.stdout + r.stderr
    finally:
        os.remove(exe)


def test_cobyla_wavefront_kernel_on_lockstep_cpu_threads_is_the_host_algorithm():
    """hip/cobyla_kernels.hip (one wavefront per LN_COBYLA search: independent sums one per lane, deciding sums by every lane, rotation chains
    one row of Z per lane) compiled by g++ over tools/simt_emu — 64 threads, barriers where the kernel has them — against the host
    algorithm (cobyla_host.c, itself the reference's run evaluation by evaluation: tests/test_cobyla_differential.py): result code,
    evaluation count, f and minimiser identical for starts inside the box, on its bounds, with infinite bounds and with a caller's step
    (tools/cobyla_emu_check.py tiny).  The device run against the real reference: tests/test_gpu_cobyla.py."""
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not shutil.which("g++") or not os.path.exists(os.path.join(root, "oracle", "libnlopt_amd_emu.so")):
        pytest.skip("no g++ / no emulated library here")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "cobyla_emu_check.py"), "tiny"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "cobyla emu check: ok" in r.stdout and "DIFFERENT" not in r.stdout, r.stdout + r.stderr


def test_ordered_set_and_list_of_worst_rows_against_a_plain_array():
    """crs_driver.c's ordered set (4-ary max-heap, keys in the nodes, batch repair) and the sorted list of worst rows the walk follows
    between two looks at the heap (redrawn beside the device) — the file is #included by tools/ordset_check.c, so these are the product's
    static functions — against the row with the largest (f, row) key of a plain array at every trial and the sorted array for every
    window's list: populations with ties everywhere, populations smaller than the window, lists that run short, redraws while a window
    is in flight, values landing among the listed rows; after a redraw the heap is a heap over every row with node key == F[row]."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    emu = os.path.join(root, "oracle", "libnlopt_amd_emu.so")
    if not shutil.which("gcc") or not os.path.exists(emu):
        pytest.skip("no gcc / no emulated library here")
    out = os.path.join(root, "tools", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "ordset_check.%d" % os.getpid())
    subprocess.run(["gcc", "-O1", "-std=gnu11", "-ffp-contract=off", "-I", os.path.join(root, "nlopt_amd", "csrc"), "-I", os.path.join(root, "include"),
                    os.path.join(root, "tools", "ordset_check.c"), "-o", exe, "-L", os.path.join(root, "oracle"), "-l:libnlopt_amd_emu.so", "-lm",
                    "-Wl,-rpath," + os.path.join(root, "oracle")], check=True)
    try:
        for seed in ("1", "2", "3"):
            r = subprocess.run([exe, "250", seed], capture_output=True, text=True, timeout=600)
            assert r.returncode == 0 and r.stdout.startswith("ok 250"), r.stdout + r.stderr
    finally:
        os.remove(exe)


def test_stochastic_ranking_kernels_in_lockstep_emulation():
    """hip/isres_stochrank.h — isres_stochrank_kernel compiled by g++ with the wavefront primitives emulated: the 64 lanes of a unit
    are threads in lockstep (DPP wave shifts, v_readfirstlane), all units of the pipeline run at once, some of them slowed down.  The
    kernel reproduces the reference's double loop (isres.c:206-228): final order, per-sweep "swapped" flags, every unit's counter at
    pop (tools/stochrank_check.cpp)."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not shutil.which("g++"):
        pytest.skip("no g++ here")
    out = os.path.join(root, "tools", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "stochrank_check.%d" % os.getpid())
    subprocess.run(["g++", "-O1", "-std=c++17", "-pthread", "-I", os.path.join(root, "nlopt_amd", "csrc", "hip"),
                    os.path.join(root, "tools", "stochrank_check.cpp"), "-o", exe], check=True)
    try:
        r = subprocess.run([exe, "5"], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout + r.stderr
    finally:
        os.remove(exe)
