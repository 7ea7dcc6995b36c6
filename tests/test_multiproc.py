"""world_size-2 tests on CPU (gloo): the collective layer of the multi-GPU runs (nlopt_amd/csrc/comm.c) and the
product's CRS driver with the initial population produced in rank blocks and all-gathered (over the CPU
emulation of the device engine — no GPU here; the device version of the same paths is tests/test_gpu_multiproc.py)."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

import _oracle as O
from _mp_launch import run_world

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_comm_partition_and_allgather_world2():
    res = run_world("comm", world=2)
    for r, d in enumerate(res):
        for count, per, first, mine in d["cover"]:
            assert per == (count + 1) // 2
            assert first == min(per * r, count) and mine == min(per, count - first)
        assert np.array_equal(d["gathered"], np.array([np.arange(5.0), np.arange(5.0) + 100.0]))
        assert d["big_ok"][0] == 1
        assert d["counters"][0] == 2 and d["counters"][1] == 2 * (40 + 300001)
    # the two ranks' blocks tile [0, count)
    for i in range(len(res[0]["cover"])):
        c0, c1 = res[0]["cover"][i], res[1]["cover"][i]
        assert c0[2] == 0 and c0[2] + c0[3] == c1[2] and c1[2] + c1[3] == c0[0]


def test_comm_world3_uneven():
    res = run_world("comm", world=3)
    for i, count in enumerate((1, 2, 7, 64, 1001)):
        blocks = [(d["cover"][i][2], d["cover"][i][3]) for d in res]
        assert blocks[0][0] == 0 and sum(m for _, m in blocks) == count
        for a, b in zip(blocks, blocks[1:]):
            assert a[0] + a[1] == b[0]


@pytest.mark.parametrize("obj,n,pop,seed,maxeval", [("rastrigin", 8, 61, 7, 1500), ("griewank", 5, 0, 3, 900)])
def test_crs_driver_sharded_init_world2_matches_oracle(obj, n, pop, seed, maxeval):
    """every rank: identical run, identical to the single-process oracle, evaluation by evaluation"""
    res = run_world("emu_crs", dict(obj=obj, n=n, pop=pop, seed=seed, maxeval=maxeval), world=2)
    p = O.run_port_crs(obj, n, pop, seed, maxeval=maxeval, trace_cap=maxeval + 64)
    for d in res:
        assert d["ret"][0] == p["ret"] and d["nevals"][0] == p["nevals"] and d["words"][0] == p["words"]
        assert np.array_equal(d["row"], p["trace"]["row"]) and np.array_equal(d["accepted"], p["trace"]["accepted"])
        assert np.array_equal(d["f"], p["trace"]["f"]) and np.array_equal(d["x"], p["x"]) and d["minf"][0] == p["minf"]
        assert d["collectives"][0] == 2         # rows + f, once (the engine emulation of oracle/port_emu_engine.c has no set-up exchange)


# ---- the product's ISRES / MLSL host drivers over the CPU stand-in for the device layer (oracle/emu_device.c) -------------------
EMU = dict(NLA_TEST_EMU_DEVICE="1")


def _check_against_oracle(d, p, nsamp_kind=None):
    assert d["ret"][0] == p["ret"] and d["nevals"][0] == p["nevals"]
    assert d["minf"][0] == p["minf"] and np.array_equal(d["x"], p["x"])


@pytest.mark.parametrize("world", [1, 2, 3])
@pytest.mark.parametrize("obj,n,pop,seed,ncon,gens", [("rastrigin", 12, 60, 5, 2, 6), ("griewank", 7, 0, 11, 0, 3), ("ackley", 20, 45, 2, 3, 5)])
def test_isres_driver_over_emulated_device_matches_oracle(world, obj, n, pop, seed, ncon, gens):
    """ISRES, evaluation and ranking-bit generation sharded over the ranks + 4 (+1: the bits) all-gathers per generation (and one small one in which the ranks agree on
    the clock / force_stop verdict of the generation, comm.c nla_comm_agree_stop): every rank reproduces the oracle's evaluation
    sequence (f and penalty of every candidate, bit for bit), result and stream position"""
    effpop = pop or 20 * (n + 1)
    a = dict(obj=obj, n=n, pop=pop, seed=seed, maxeval=gens * effpop, ncon=ncon)
    p = O.run_port_isres(obj, n, pop, seed, nineq=ncon, maxeval=gens * effpop)
    for d in run_world("gpu_isres", a, world=world, extra_env=EMU):
        _check_against_oracle(d, p)
        assert np.array_equal(d["f"], p["ftrace"][:len(d["f"])]) and len(d["f"]) == len(p["ftrace"])
        if world > 1:
            # + one all-gather of the ranking bits in every generation that ranks stochastically (some infeasible individual)
            c = d["collectives"][0] - 1            # (the set-up's "ready" exchange)
            assert 5 * gens <= c <= 6 * gens and (ncon > 0 or c == 5 * gens)


def test_isres_generator_state_array_sized_up_front_over_the_emulated_device():
    """a population large enough for nla_mtstream_expect (mtstream.c) to re-allocate the generator's segment-state array at set-up —
    16 generations of 2 (pop - 1) pop ranking uniforms are more than the 64 states it starts with (round 5: the array's doublings used
    to free device memory in the middle of generations) — and the run is still the oracle's, evaluation by evaluation"""
    obj, n, pop, seed, ncon, gens = "rastrigin", 3, 1300, 4, 1, 2
    a = dict(obj=obj, n=n, pop=pop, seed=seed, maxeval=gens * pop, ncon=ncon)
    p = O.run_port_isres(obj, n, pop, seed, nineq=ncon, maxeval=gens * pop)
    for d in run_world("gpu_isres", a, world=1, extra_env=EMU):
        _check_against_oracle(d, p)
        assert np.array_equal(d["f"], p["ftrace"][:len(d["f"])]) and len(d["f"]) == len(p["ftrace"])


@pytest.mark.parametrize("world", [1, 2])
@pytest.mark.parametrize("obj,n,ns,seed,local,lds,maxeval", [
    ("rastrigin", 5, 12, 5, "lbfgs", False, 1500), ("griewank", 6, 40, 3, "lbfgs", False, 3000), ("ackley", 8, 0, 9, "mma", False, 2500),
    ("levy", 4, 16, 7, "mma", True, 1200), ("rosenbrock", 4, 12, 9, "lbfgs", True, 2000), ("rastrigin", 6, 20, 11, "default", False, 3000),
    ("sphere", 5, 6, 2, "default", True, 400),
    ("ackley", 40, 500, 6, "lbfgs", False, 1600)])      # (n N large enough for the generator's state array to be re-allocated at set-up, nla_mtstream_expect)
def test_mlsl_driver_over_emulated_device_matches_oracle(world, obj, n, ns, seed, local, lds, maxeval):
    """MLSL with LD_LBFGS / LD_MMA, pseudo-random and Sobol sampling, local searches dealt over the ranks and all-gathered: the
    batched, speculative walk of mlsl_driver.c commits exactly what the serial reference commits — same samples, same local
    minima with the same evaluation counts, in the same order"""
    a = dict(obj=obj, n=n, pop=ns, seed=seed, maxeval=maxeval, local=local, lds=lds)
    p = O.run_port_mlsl(obj, n, ns, seed, maxeval=maxeval, local="mma" if local == "default" else local, lds=lds)
    for d in run_world("gpu_mlsl", a, world=world, extra_env=EMU):
        _check_against_oracle(d, p)
        samp, loc = d["kind"] == 3, d["kind"] == 4
        assert np.array_equal(d["f"][samp], p["fsamp"][:samp.sum()]) and samp.sum() in (len(p["fsamp"]), len(p["fsamp"]) - 1)
        assert np.array_equal(d["f"][loc], p["floc"]) and np.array_equal(d["accepted"][loc], p["eloc"])


@pytest.mark.parametrize("world,first,env", [(1, 0, {}), (1, 15, {"NLA_CRS_UPLOAD": "1", "NLA_CRS_COPY_STATUS": "1"}), (2, 30, {}), (3, 40, {}),
                                             (1, 50, {"NLA_EMU_EVOLVE2": "1"}), (2, 65, {"NLA_EMU_EVOLVE2": "1"})])
def test_drawn_configurations_of_the_host_drivers_over_the_emulated_device(world, first, env):
    """CRS2_LM (whole product path incl. crs_engine.c: window factor, speculation cap, host-callback mode, the alternative list /
    status transports), ESCH, ISRES (with NLA_EMU_EVOLVE2 the emulated device plays the multi-start evolve's protocol — rounds of
    at most 256 individuals, forced hand-overs to the serial kernel, deviates running out mid-round — so the driver's round / refill
    / fallback loop runs) and MLSL: objective, dimension, population / samples, seed, constraints, stop value, local optimiser + tolerance + its own evaluation
    limit, Sobol or pseudo-random sampling — drawn; each run compared with the oracle inside the worker (result, evaluation count,
    every candidate / local minimum, position of the generator afterwards)"""
    count = 15 if world == 1 else 10
    for d in run_world("emu_sweep", dict(first=first, count=count), world=world, extra_env=dict(EMU, **env), timeout=900):
        assert d["checked"][0] == count


# ---- CRS2_LM with the population sharded BY COORDINATE over the ranks (hip/crs_shard.hip, crs_engine.c) ---------------------------
SHARDED_CRS = [
    # obj, n, pop, seed, maxeval, extra
    ("rastrigin", 10, 100, 42, 1400, {}),                       # BASELINE config 1's shape
    ("rosenbrock", 7, 40, 9, 900, dict(xtol_rel=1e-3)),          # coordinates coupled across slice boundaries; x needed on the host (xtol)
    ("levy", 9, 0, 12345, 1200, {}),                             # head term wants x_0 and x_{n-1} (first and last rank)
    ("griewank", 257, 600, 5, 1100, {}),                         # odd n: unequal slices (world 2: 129 + 128, world 3: 86 + 86 + 85)
    ("ackley", 64, 300, 7, 1500, dict(params={"amd_window_factor": 3.0})),
    ("sphere", 5, 30, 3, 600, dict(ftol_rel=1e-6)),
]


def _no_windows(extra):
    """the conservative passes + one all-gather per pass (round 3's sharded path; since round 6 the fallback where the coordinates cannot
    be dealt in whole cache lines, where a rank cannot map its peers' buffers, or on request)"""
    return dict(extra, params=dict(extra.get("params") or {}, amd_shard_windows=0))


SHARDED_WINDOWS = [
    # world, obj, n, pop, seed, maxeval, extra
    (2, "ackley", 64, 300, 7, 1500, {}),                                   # 32 + 32 columns
    (2, "griewank", 257, 600, 5, 1100, {}),                                # 144 + 113 columns, odd n
    (3, "griewank", 257, 600, 5, 1100, {}),                                # 96 + 96 + 65
    (3, "levy", 130, 400, 12345, 1300, {}),                                # coupled across slice boundaries (48 + 48 + 34)
    (2, "rosenbrock", 48, 200, 9, 2500, dict(xtol_rel=1e-3)),              # x needed on the host: whole points are local in this mode
    (4, "rastrigin", 128, 2000, 42, 4000, dict(params={"amd_window_factor": 3.0})),
    (2, "sphere", 40, 300, 3, 4000, dict(ftol_rel=1e-6)),
]


@pytest.mark.parametrize("world,obj,n,pop,seed,maxeval,extra", SHARDED_WINDOWS)
def test_crs_column_sharded_windows_are_the_oracles_run(world, obj, n, pop, seed, maxeval, extra):
    """round 6: the column-sharded population with the window resolved on the device — over the emulated device every rank's launcher forms
    its columns of a slot, stores them into every rank's TX (shared memory mapped through the library's nla_ipc_* layer, as the HIP build
    maps peer device memory), raises its chunk flags and waits for the other PROCESSES' flags before it evaluates: the engine's set-up
    (handle exchange, table), the per-window protocol (launch numbers, flags that are never cleared, stop words) and the driver above it
    are the product's.  The run is the single-process oracle's bit for bit on every rank, with no collective per window."""
    kw = {k: v for k, v in extra.items() if k in ("ftol_rel", "xtol_rel")}
    res = run_world("gpu_crs", dict(obj=obj, n=n, pop=pop, seed=seed, maxeval=maxeval, **extra), world=world, extra_env=EMU)
    p = O.run_port_crs(obj, n, pop, seed, maxeval=maxeval, trace_cap=maxeval + 4096, **kw)
    for d in res:
        assert d["ret"][0] == p["ret"] and d["nevals"][0] == p["nevals"] and d["minf"][0] == p["minf"]
        assert np.array_equal(d["x"], p["x"])
        for key in ("f", "row", "kind", "accepted"):
            assert np.array_equal(d[key], p["trace"][key]), key
        assert d["after"][0] == res[0]["after"][0]
        assert d["collectives"][0] <= 12 and d["rounds"][0] >= 3, (d["collectives"][0], d["rounds"][0])       # set-up exchanges only
        assert d["stats_allgather_bytes"][0] > 0


def test_crs_column_sharded_windows_fall_back_together_when_a_peer_cannot_be_mapped():
    """NLA_EMU_NO_IPC: exporting the window buffers fails (on every rank here; one failing rank takes the same path: the handle exchange
    carries an invalid handle) -> all ranks agree and run the conservative passes: same run, a collective per pass"""
    a = dict(obj="ackley", n=64, pop=300, seed=7, maxeval=1500)
    res = run_world("gpu_crs", a, world=2, extra_env=dict(EMU, NLA_EMU_NO_IPC="1"))
    p = O.run_port_crs("ackley", 64, 300, 7, maxeval=1500, trace_cap=6000)
    for d in res:
        assert d["ret"][0] == p["ret"] and np.array_equal(d["f"], p["trace"]["f"]) and np.array_equal(d["x"], p["x"])
        assert d["collectives"][0] >= d["rounds"][0]


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("obj,n,pop,seed,maxeval,extra", [(o_, n_, p_, s_, m_, _no_windows(e_)) for o_, n_, p_, s_, m_, e_ in SHARDED_CRS])
def test_crs_column_sharded_over_the_ranks_is_the_oracles_run(world, obj, n, pop, seed, maxeval, extra):
    """every rank keeps 1/world of the COLUMNS of the population, runs the gather-sum, mutation and row replacement on its slice, the
    candidates of a pass are all-gathered and evaluated by every rank: the run — every f, every decision, the result, the stream
    position — is the single-process oracle's, bit for bit, on every rank"""
    kw = {k: v for k, v in extra.items() if k in ("ftol_rel", "xtol_rel")}
    res = run_world("gpu_crs", dict(obj=obj, n=n, pop=pop, seed=seed, maxeval=maxeval, **extra), world=world, extra_env=EMU)
    p = O.run_port_crs(obj, n, pop, seed, maxeval=maxeval, trace_cap=maxeval + 4096, **kw)
    for d in res:
        assert d["ret"][0] == p["ret"] and d["nevals"][0] == p["nevals"] and d["minf"][0] == p["minf"]
        assert np.array_equal(d["x"], p["x"])
        for key in ("f", "row", "kind", "accepted"):
            assert np.array_equal(d[key], p["trace"][key]), key
        assert d["after"][0] == res[0]["after"][0]
        # sharded for real: an all-gather of candidates per pass (+ the initial values, + the stop agreements)
        assert d["collectives"][0] >= d["rounds"][0] and d["stats_allgather_bytes"][0] > 0


def test_crs_column_sharding_can_be_switched_off_and_falls_back_to_replicas():
    res = run_world("gpu_crs", dict(obj="rastrigin", n=10, pop=100, seed=42, maxeval=700, params={"amd_shard": 0}), world=2, extra_env=EMU)
    p = O.run_port_crs("rastrigin", 10, 100, 42, maxeval=700, trace_cap=5000)
    for d in res:
        assert np.array_equal(d["f"], p["trace"]["f"]) and np.array_equal(d["x"], p["x"])
        assert d["collectives"][0] == 4           # the row-sharded initialisation only: rows + values (+ the set-up's two "ready" exchanges)
    # n < world columns cannot be dealt: replicas as well
    res = run_world("gpu_crs", dict(obj="sphere", n=2, pop=20, seed=1, maxeval=300), world=3, extra_env=EMU)
    p = O.run_port_crs("sphere", 2, 20, 1, maxeval=300, trace_cap=5000)
    for d in res:
        assert np.array_equal(d["f"], p["trace"]["f"]) and np.array_equal(d["x"], p["x"])


def test_crs_column_sharded_large_n_prefix_world3():
    """above the dimension where a single GPU switches to the chain kernel (n >= 2048; sharded runs stay with conservative passes):
    n = 2050 over 3 ranks (684 + 684 + 682 columns), a prefix of the trial chain.  (The device twin, tests/test_gpu_multiproc.py, runs
    the metric's own n = 4096; here that costs 40 s of emulation.)"""
    n, pop, seed, me = 2050, 2100, 42, 2230
    res = run_world("gpu_crs", dict(obj="griewank", n=n, pop=pop, seed=seed, maxeval=me, params={"amd_shard_windows": 0}), world=3, extra_env=EMU, timeout=1200)
    p = O.run_port_crs("griewank", n, pop, seed, maxeval=me, trace_cap=me + 64)
    for d in res:
        assert d["ret"][0] == p["ret"] and d["nevals"][0] == p["nevals"]
        for key in ("f", "row", "kind", "accepted"):
            assert np.array_equal(d[key], p["trace"][key]), key
        assert np.array_equal(d["x"], p["x"])


def _single_emu(a):
    """the same job in ONE process over the emulated device (what every rank of a sharded run must reproduce)"""
    return run_world("gpu_crs", dict(a, sharded=False), world=1, extra_env=EMU)[0]


@pytest.mark.parametrize("a", [
    dict(obj="rastrigin", n=8, pop=60, seed=5, maxeval=900, maximize=True),                 # nlopt_set_max_objective: the kernels deliver -f on every rank
    dict(obj="sphere", n=6, pop=40, seed=3, maxeval=5000, stopval=0.05),                     # STOPVAL_REACHED mid-run
    dict(obj="ackley", n=12, pop=80, seed=9, maxeval=700, fix_last=True),                    # a fixed coordinate: host wrapper -> replicas, still identical
    dict(obj="griewank", n=9, pop=50, seed=2, maxeval=500, twice=True),                      # the object reused: the generator continues
], ids=["maximize", "stopval", "fixed_dim", "twice"])
def test_crs_column_sharded_variants_equal_the_single_process_run(a):
    s = _single_emu(a)
    for d in run_world("gpu_crs", a, world=2, extra_env=EMU):
        assert d["ret"][0] == s["ret"][0] and d["nevals"][0] == s["nevals"][0] and d["minf"][0] == s["minf"][0]
        assert np.array_equal(d["x"], s["x"]) and np.array_equal(d["f"], s["f"]) and np.array_equal(d["row"], s["row"])
        assert d["after"][0] == s["after"][0]
        if a.get("twice"):
            assert np.array_equal(d["x1"], s["x1"]) and d["minf1"][0] == s["minf1"][0] and d["nevals1"][0] == s["nevals1"][0]
            assert not np.array_equal(d["x1"], d["x"])


@pytest.mark.parametrize("windows", [0, 1], ids=["passes", "windows"])
@pytest.mark.parametrize("a", [dict(force_stop_rank=1, force_stop_after=0.3), dict(maxtime_rank=0, maxtime=4.0)], ids=["force_stop_on_one_rank", "maxtime_on_one_rank"])
def test_crs_column_sharded_ranks_leave_together(a, windows):
    """a stop condition only ONE rank sees (its user's force_stop, its own clock) is agreed by all ranks once per pass: every rank
    returns the same result after the same number of evaluations — nobody is left waiting in an all-gather"""
    cfg = dict(obj="rastrigin", n=64, pop=400, seed=7, maxeval=2000000, params={"amd_shard_windows": windows}, **a)      # (windows: the stop bits cross inside the launch)
    res = run_world("gpu_crs", cfg, world=2, extra_env=EMU, timeout=300)
    want = -5 if "force_stop_rank" in a else 6                  # NLOPT_FORCED_STOP / NLOPT_MAXTIME_REACHED
    assert res[0]["ret"][0] == want and res[1]["ret"][0] == want
    assert res[0]["nevals"][0] == res[1]["nevals"][0] and 400 < res[0]["nevals"][0] < 2000000
    assert np.array_equal(res[0]["x"], res[1]["x"]) and res[0]["minf"][0] == res[1]["minf"][0]


@pytest.mark.parametrize("world,fail_rank,transport", [(2, 1, "gloo"), (3, 0, "gloo"), (2, 0, "shm"), (3, 2, "shm")])
def test_crs_column_sharded_rank_that_fails_in_a_pass_takes_the_others_with_it(world, fail_rank, transport):
    """ONE rank fails in the middle of a sharded pass (injected after its set-up was agreed: debug switch of the emulated-device build).
    It still packs and joins the pass's all-gather with the failure value in its flag word, so every rank sees it in the pass's own
    exchange and returns NLOPT_FAILURE after the same pass — nobody is left waiting in an all-gather (this test would time out)"""
    cfg = dict(obj="rastrigin", n=16, pop=200, seed=7, maxeval=4000, want_errmsg=1)
    env = dict(EMU, NLA_CRS_FAIL_RANK=str(fail_rank), NLA_CRS_FAIL_PASS="6")
    if transport == "shm":
        env["NLA_TEST_SHM"] = "1"
    res = run_world("gpu_crs", cfg, world=world, extra_env=env, timeout=300)
    assert all(d["ret"][0] == -1 for d in res), [int(d["ret"][0]) for d in res]
    assert len({int(d["nevals"][0]) for d in res}) == 1 and 200 < res[0]["nevals"][0] < 4000
    for r, d in enumerate(res):
        msg = str(d["errmsg"])
        assert ("injected failure" in msg) if r == fail_rank else ("another rank failed" in msg), (r, msg)


# ---- the RCCL transport itself with several ranks: comm.c's ncclAllGather branch over a mock librccl (oracle/mock_rccl.c) -----------
MOCK = dict(NLA_TEST_EMU_DEVICE="1", NLA_TEST_MOCK_RCCL="1")


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case,a", [
    ("gpu_crs", dict(obj="rastrigin", n=10, pop=100, seed=42, maxeval=1400)),                                   # column-sharded: candidates per pass, initial f in place
    ("gpu_crs", dict(obj="levy", n=9, pop=0, seed=12345, maxeval=1200, xtol_rel=1e-4)),                         # + whole points gathered for the host (xtol)
    ("gpu_crs", dict(obj="griewank", n=8, pop=61, seed=7, maxeval=900, params={"amd_shard": 0})),               # replicas: rows and f all-gathered IN PLACE
    ("gpu_isres", dict(obj="rastrigin", n=12, pop=60, seed=5, maxeval=6 * 60, ncon=2)),                         # f / penalties in place, ranking bits in place, stop agreement (host data)
    ("gpu_mlsl", dict(obj="ackley", n=6, pop=25, seed=7, maxeval=2500)),                                        # minimisers of a batch, distance minima (host data)
], ids=["crs_sharded", "crs_sharded_xtol", "crs_replicas_inplace", "isres", "mlsl"])
def test_rccl_transport_with_several_ranks_over_the_mock(world, case, a):
    """comm.c's RCCL branch (ncclCommInitRank, ncclAllGather on device buffers incl. the in-place form, host data staged through
    device buffers) has only ever run with ONE rank on hardware (one GPU per box).  Here it runs with 2-3 ranks: librccl is replaced by
    a mock that moves the bytes through shared memory and enforces NCCL's contract — same byte count on every rank, same collective
    order, an overlapping send buffer exactly at recvbuff + rank * count — over the emulated device.  Every rank must reproduce the
    single-process run."""
    if world == 3 and (a.get("xtol_rel") or case == "gpu_mlsl" or (a.get("params") or {}).get("amd_shard") == 0):
        pytest.skip("three ranks: the sharded CRS2_LM and the ISRES case are enough (suite time)")
    s = run_world(case, dict(a, sharded=False), world=1, extra_env=EMU)[0]
    for d in run_world(case, a, world=world, extra_env=MOCK):
        assert d["ret"][0] == s["ret"][0] and d["nevals"][0] == s["nevals"][0] and d["minf"][0] == s["minf"][0]
        assert np.array_equal(d["x"], s["x"]) and np.array_equal(d["f"], s["f"]) and np.array_equal(d["row"], s["row"])
        assert d["after"][0] == s["after"][0]
        assert d["collectives"][0] >= 2
        calls, inplace, nbytes = d["rccl_calls"]
        assert calls == d["collectives"][0] and nbytes > 0, "the run's collectives did not go through ncclAllGather"
        if case != "gpu_mlsl":
            assert inplace >= 1                                   # the initial f values (CRS) / f and penalties (ISRES) are gathered in place


def _mock_rank(path, uid, rank, world, count, misplace, q):
    import ctypes as C
    L = C.CDLL(path)
    class Uid(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
    L.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    u = Uid(); C.memmove(C.byref(u), uid, 128)
    comm = C.c_void_p()
    assert L.ncclCommInitRank(C.byref(comm), world, u, rank) == 0
    recv = np.zeros(world * 64, dtype=np.uint8)
    send = np.full(64, rank + 1, dtype=np.uint8)
    if misplace:                                         # "in place" at the wrong offset
        rc = L.ncclAllGather(recv.ctypes.data + ((rank + 1) % world) * count + 1, recv.ctypes.data, count, 1, comm, None)
    else:
        rc = L.ncclAllGather(send.ctypes.data, recv.ctypes.data, count, 1, comm, None)
    q.put((rank, rc, recv[: world * count].tolist()))
    if rc == 0:
        L.ncclCommDestroy(comm)


@pytest.mark.parametrize("what", ["agree", "counts_differ", "misplaced_in_place"])
def test_the_mock_rccl_enforces_the_allgather_contract(what):
    """the checker checks: ranks that agree get every rank's bytes rank-major; ranks that pass different counts, or an overlapping send
    buffer that is not recvbuff + rank * count, get ncclInvalidUsage (5) instead of silently wrong data"""
    import ctypes as C
    import multiprocessing as mp
    path = os.path.join(ROOT, "oracle", "libmockrccl.so")
    uid = C.create_string_buffer(128)
    assert C.CDLL(path).ncclGetUniqueId(uid) == 0
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    ps = [ctx.Process(target=_mock_rank, args=(path, uid.raw, r, 2, 8 if (what != "counts_differ" or r == 0) else 16,
                                               what == "misplaced_in_place", q)) for r in range(2)]
    for p in ps: p.start()
    got = sorted(q.get(timeout=60) for _ in ps)
    for p in ps: p.join(30)
    if what == "agree":
        assert [g[1] for g in got] == [0, 0] and got[0][2] == got[1][2] == [1] * 8 + [2] * 8
    else:
        assert all(g[1] == 5 for g in got)
    try:
        os.unlink("/dev/shm" + uid.value.decode())
    except OSError:
        pass


# ---- one rank's set-up fails: nobody is left waiting ----------------------------------------------------------------------------------
@pytest.mark.parametrize("alg,world,fail_rank", [("crs", 2, 1), ("crs", 3, 0), ("crs_replicas", 2, 0), ("isres", 2, 1), ("mlsl", 2, 1), ("mlsl", 3, 2)])
def test_a_rank_whose_setup_fails_takes_the_others_with_it(alg, world, fail_rank):
    """every device / pinned allocation of ONE rank's set-up is made to fail in turn (emulated device layer).  Without an agreement at the
    end of set-up the other ranks would enter the run's first all-gather and wait for ever (this test would time out); with it
    (comm.c, nla_comm_agree_ready) every rank returns an error — the failing one OUT_OF_MEMORY, the others FAILURE naming the reason —
    and the communicator serves the next run, which equals the first"""
    a = dict(alg=alg, n=8, pop=40, maxeval=500, seed=3, fail_rank=fail_rank)
    out = run_world("fault_setup", a, world=world, extra_env=EMU, timeout=600)
    nset = int(out[0]["nset"][0])
    assert nset >= 8 and all(int(d["nready"][0]) >= 1 for d in out), (nset, [int(d["nready"][0]) for d in out])
    for r, d in enumerate(out):
        assert d["base_ret"][0] > 0 and d["again_ret"][0] == d["base_ret"][0]
        assert d["again_minf"][0] == d["base_minf"][0] and np.array_equal(d["again_x"], d["base_x"]) and d["again_nevals"][0] == d["base_nevals"][0]
        assert d["base_minf"][0] == out[0]["base_minf"][0] and np.array_equal(d["base_x"], out[0]["base_x"])
        assert len(d["rets"]) == nset
        assert d["live"][0] == 0, "rank %d: %d device-layer objects left behind by the failed set-ups" % (r, d["live"][0])
    for k in range(nset):
        rets = [int(d["rets"][k]) for d in out]
        assert all(x < 0 for x in rets), "allocation %d of rank %d failed: results %r" % (k + 1, fail_rank, rets)
        assert rets[fail_rank] in (-3, -1)
        for r, d in enumerate(out):
            assert str(d["msgs"][k]), (k, r)
            if r != fail_rank:
                assert rets[r] == -1 and "another rank" in str(d["msgs"][k]), (k, r, rets, str(d["msgs"][k]))


@pytest.mark.parametrize("case,a", [
    ("gpu_crs", dict(obj="rastrigin", n=10, pop=100, seed=42, maxeval=1400)),
    ("gpu_crs", dict(obj="rastrigin", n=10, pop=100, seed=42, maxeval=1400, params={"amd_shard": 0})),
    ("gpu_isres", dict(obj="rastrigin", n=12, pop=60, seed=5, maxeval=360, ncon=2)),
    ("gpu_mlsl", dict(obj="ackley", n=6, pop=25, seed=7, maxeval=2500)),
], ids=["crs_sharded", "crs_replicas", "isres", "mlsl"])
@pytest.mark.parametrize("mistake", ["seed_by_rank", "x0_by_rank", "param_by_rank"])
def test_ranks_given_different_jobs_are_told_so(case, a, mistake):
    if mistake != "seed_by_rank" and not (case == "gpu_crs" and "params" not in a):
        pytest.skip("the starting point's share of the fingerprint: one algorithm is enough (suite time)")
    """one job over several ranks needs the identical problem and generator state on every rank; ranks seeded differently (the classic
    mistake: seed = base + rank) or started from different points would take different decisions and pass each other in the
    collectives — and so would ranks whose run-shaping options differ (param_by_rank: "amd_window_factor" sets the window depth, hence
    the size of every pass's all-gather).  The set-up's exchange carries a fingerprint of the job: every rank returns NLOPT_INVALID_ARGS
    and says why."""
    for d in run_world(case, dict(a, want_errmsg=True, **{mistake: True}), world=2, extra_env=EMU, timeout=120):
        assert d["ret"][0] == -2 and d["nevals"][0] == 0
        assert "different problems" in str(d["errmsg"]) and "nlopt_srand" in str(d["errmsg"])


@pytest.mark.parametrize("world,env,obj,n,pop,seed,ncon,gens", [
    (1, {}, "rastrigin", 12, 60, 5, 2, 6), (1, {}, "sphere", 6, 30, 5, 1, 40), (1, {"NLA_EMU_EVOLVE2": "1"}, "ackley", 20, 45, 2, 3, 5),
    (2, {"NLA_EMU_EVOLVE2": "1"}, "sphere", 6, 30, 5, 1, 40), (3, {}, "rastrigin", 12, 60, 5, 2, 6)])
def test_isres_overlap_mode_changes_nothing(world, env, obj, n, pop, seed, ncon, gens):
    """ISRES with "amd_isres_overlap" = 1 (the default since it was measured; = 0 also run here, isres_driver.c): the generator works on a second stream — ranking bits beside the
    rank counting, the evolve phase's deviates generated AHEAD beside the ranking pipeline (thrown away when the ranking stops
    early: the sphere case does, late in its run), the next ranking's segment states beside the evolve rounds.  The bookkeeping
    (stream positions of the speculation, what is reused and what is redone) must leave every candidate, the result and the
    generator where the oracle has them.  (The emulated device is synchronous: what this cannot see is a missing synchronisation.)"""
    p = O.run_port_isres(obj, n, pop, seed, nineq=ncon, maxeval=gens * pop)
    for ov in ((1, 0) if world == 1 else (1,)):
        a = dict(obj=obj, n=n, pop=pop, seed=seed, maxeval=gens * pop, ncon=ncon, params={"amd_isres_overlap": ov})
        for d in run_world("gpu_isres", a, world=world, extra_env=dict(EMU, **env)):
            _check_against_oracle(d, p)
            assert np.array_equal(d["f"], p["ftrace"][:len(d["f"])]) and len(d["f"]) == len(p["ftrace"])


@pytest.mark.parametrize("world", [1, 2])
@pytest.mark.parametrize("obj,n,ns,seed,local,maxeval", [("rastrigin", 5, 12, 5, "lbfgs", 1500), ("griewank", 6, 40, 3, "lbfgs", 3000),
                                                          ("ackley", 8, 0, 9, "mma", 2500), ("rastrigin", 6, 20, 11, "default", 3000)])
def test_mlsl_prefetch_of_the_next_samples_changes_nothing(world, obj, n, ns, seed, local, maxeval):
    """MLSL's sample prefetch (mlsl_driver.c, always on with a device objective): pseudo-random sampling's stream words of the NEXT iteration are
    generated on a second stream while the distance pass and the local searches of this one run (nothing in between draws random
    numbers).  Same samples, same local searches, same final generator position as the oracle — incl. the run's last iteration, whose
    prefetched words are never used.  (The emulated device is synchronous: what this cannot see is a missing synchronisation.)"""
    a = dict(obj=obj, n=n, pop=ns, seed=seed, maxeval=maxeval, local=local, lds=False)
    p = O.run_port_mlsl(obj, n, ns, seed, maxeval=maxeval, local="mma" if local == "default" else local, lds=False)
    for d in run_world("gpu_mlsl", a, world=world, extra_env=EMU):
        _check_against_oracle(d, p)
        samp, loc = d["kind"] == 3, d["kind"] == 4
        assert np.array_equal(d["f"][samp], p["fsamp"][:samp.sum()]) and samp.sum() in (len(p["fsamp"]), len(p["fsamp"]) - 1)
        assert np.array_equal(d["f"][loc], p["floc"]) and np.array_equal(d["accepted"][loc], p["eloc"])


# ---- the library's shared-memory transport (comm.c, nlopt_amd_comm_create_shm): no Python in the exchange ---------------------------
@pytest.mark.parametrize("world,slot", [(2, 0), (3, 0), (3, 4096)])
def test_shm_transport_allgather(world, slot):
    """the same contract as the gloo transport's (partition, rank-major all-gather, counters) — with a 4 KB slot the 300 001-byte
    payload travels in 74 pieces through the two alternating slot sets"""
    res = run_world("comm", world=world, extra_env=dict(NLA_TEST_SHM="1", NLA_TEST_SHM_SLOT=str(slot)))
    for r, d in enumerate(res):
        assert np.array_equal(d["gathered"], np.array([np.arange(5.0) + 100.0 * q for q in range(world)]))
        assert d["big_ok"][0] == 1
        assert d["counters"][0] == 2 and d["counters"][1] == world * (40 + 300001)


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("obj,n,pop,seed,maxeval,extra", [SHARDED_CRS[0][:5] + (_no_windows(SHARDED_CRS[0][5]),), SHARDED_CRS[3][:5] + (_no_windows(SHARDED_CRS[3][5]),)])
def test_crs_column_sharded_over_the_shm_transport(world, obj, n, pop, seed, maxeval, extra):
    """the column-sharded CRS2_LM run with the candidates exchanged through the shared-memory transport: the oracle's run bit for bit"""
    kw = {k: v for k, v in extra.items() if k in ("ftol_rel", "xtol_rel")}
    res = run_world("gpu_crs", dict(obj=obj, n=n, pop=pop, seed=seed, maxeval=maxeval, **extra), world=world, extra_env=dict(EMU, NLA_TEST_SHM="1"))
    p = O.run_port_crs(obj, n, pop, seed, maxeval=maxeval, trace_cap=maxeval + 4096, **kw)
    for d in res:
        assert d["ret"][0] == p["ret"] and d["nevals"][0] == p["nevals"] and d["minf"][0] == p["minf"] and np.array_equal(d["x"], p["x"])
        for key in ("f", "row", "kind", "accepted"):
            assert np.array_equal(d[key], p["trace"][key]), key
        assert d["collectives"][0] >= d["rounds"][0] and d["stats_allgather_bytes"][0] > 0


def test_isres_over_the_shm_transport_world3():
    obj, n, pop, seed, ncon, gens = "rastrigin", 12, 60, 5, 2, 6
    res = run_world("gpu_isres", dict(obj=obj, n=n, pop=pop, seed=seed, maxeval=gens * pop, ncon=ncon), world=3, extra_env=dict(EMU, NLA_TEST_SHM="1", NLA_TEST_SHM_SLOT="4096"))
    p = O.run_port_isres(obj, n, pop, seed, nineq=ncon, maxeval=gens * pop)
    for d in res:
        _check_against_oracle(d, p)


SHM_START_SNIPPET = r"""
import ctypes as C, os, struct, sys, time
rank, world, name, delay, lib = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], float(sys.argv[4]), sys.argv[5]
L = C.CDLL(lib)
L.nlopt_amd_comm_create_shm.restype = C.c_void_p
L.nlopt_amd_comm_create_shm.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_size_t]
L.nlopt_amd_comm_set_timeout.argtypes = [C.c_void_p, C.c_double]
L.nla_comm_allgather_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
L.nlopt_amd_comm_destroy.argtypes = [C.c_void_p]
time.sleep(delay)
c = L.nlopt_amd_comm_create_shm(rank, world, name.encode(), 4096)
if not c:
    print("create failed"); sys.exit(3)
L.nlopt_amd_comm_set_timeout(c, 30.0)
send = struct.pack("q", 1000 + rank)
recv = C.create_string_buffer(8 * world)
rc = L.nla_comm_allgather_host(c, send, recv, 8, None)
print("rc", rc, "got", struct.unpack("%dq" % world, recv.raw))
L.nlopt_amd_comm_destroy(c)
"""


@pytest.mark.parametrize("late_rank0", [False, True])
def test_shm_transport_start_up_ignores_the_segment_of_a_crashed_run(late_rank0):
    """comm.c start-up: a segment a crashed run left under the job's name — valid magic, world and slot size, barrier words
    holding garbage, creator dead — must not catch a rank > 0 that starts before rank 0 has replaced it.  The stale segment is planted
    by hand; rank 0 arrives 0.7 s after the others (or first); every rank must finish one all-gather within seconds."""
    import struct
    world, slot = 3, 4096
    name = "/nla_test_stale_%d_%d" % (os.getpid(), int(late_rank0))
    path = "/dev/shm" + name
    total = (64 + 2 * world * slot + 4095) & ~4095
    # header layout of comm.c's shm_header: magic, world, slot (u64), arrived, generation, attached, detached, aborted, creator_pid, creator_start (u64)
    dead = subprocess.Popen([sys.executable, "-c", "pass"])
    dead.wait()
    hdr = struct.pack("<IIQIIIIIIQ", 0x6e6c6173, world, slot, 2, 7, world, 1, 0, dead.pid, 12345)
    with open(path, "wb") as fh:
        fh.write(hdr + b"\0" * (total - len(hdr)))
    try:
        procs = []
        for r in range(world):
            delay = (0.7 if r == 0 else 0.0) if late_rank0 else (0.0 if r == 0 else 0.3)
            procs.append(subprocess.Popen([sys.executable, "-c", SHM_START_SNIPPET, str(r), str(world), name, str(delay), os.path.join(ROOT, "oracle", "libnlopt_amd_emu.so")],
                                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        t0 = time.time()
        outs = [p.communicate(timeout=100) for p in procs]
        for p, (so, se) in zip(procs, outs):
            assert p.returncode == 0, so + se
            assert "rc 0 got (1000, 1001, 1002)" in so, so + se
        assert time.time() - t0 < 60
        assert not os.path.exists(path)                  # the last rank to leave removed the name
    finally:
        if os.path.exists(path):
            os.unlink(path)
