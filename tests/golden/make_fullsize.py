"""Generate the FULL-SIZE golden fixtures tests/golden/full_*.npz: BASELINE.json's configurations at the sizes the
contract names, run through the REAL reference (oracle/_ref/libnlopt_ref.so built from /root/reference by
oracle/Makefile) and — where the reference does not expose what the test needs (row indices, accept flags) or cannot
run at all (config 5: 32-bit index overflow, crs.c:101,212) — through the 64-bit port AFTER the port has been checked
against the reference evaluation by evaluation in this same script.  Build container only (minutes to an hour of CPU,
up to 33 GB of RAM):

    python tests/golden/make_fullsize.py <case> [<case> ...]        # or: all

What a fixture holds is compact (full per-evaluation arrays would be MBs): the trial-phase trace in full (rows, kinds,
accept flags, f), the initial population's f sub-sampled (every 16th row) plus its extreme order statistics, block
sums of long f sequences, the argmin x, and counts.  tests/test_gpu_fullsize.py compares the HIP path with them."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _oracle as O  # noqa: E402


def block_sums(a, b):
    a = np.asarray(a, dtype=np.float64)
    m = (len(a) // b) * b
    return a[:m].reshape(-1, b).sum(axis=1)


def save(name, **kw):
    path = os.path.join(HERE, "full_%s.npz" % name)
    np.savez_compressed(path, **kw)
    print("wrote", path, os.path.getsize(path), "bytes", flush=True)


def crs_case(name, obj, n, N, extra, seed=42, with_ref=True):
    """CRS2_LM: N initial evaluations + `extra` trial-phase evaluations.  reference: f and the hash of x of every
    evaluation; port: the same plus (row, kind, accepted).  The two are required to agree before anything is stored."""
    me = N + extra
    t0 = time.time()
    p = O.run_port_crs(obj, n, N, seed, maxeval=me, trace_cap=me + 50000, record=True)
    print(name, "port: ret", p["ret"], "nevals", p["nevals"], "minf", p["minf"], "%.0f s" % (time.time() - t0), flush=True)
    tr = p["trace"]
    assert len(tr) == p["nevals"] == len(p["fseq"])
    assert np.array_equal(tr["f"], p["fseq"])
    ref_checked = 0
    if with_ref:
        t0 = time.time()
        r = O.run_ref(19, obj, n, N, seed, maxeval=me)
        print(name, "reference: ret", r["ret"], "nevals", r["nevals"], "minf", r["minf"], "%.0f s" % (time.time() - t0), flush=True)
        assert r["ret"] == p["ret"] and r["nevals"] == p["nevals"] and r["minf"] == p["minf"]
        assert np.array_equal(r["fseq"], p["fseq"]) and np.array_equal(r["xhash"], p["xhash"])    # bit for bit, every evaluation
        assert np.array_equal(r["x"], p["x"])
        ref_checked = 1
    F0 = tr["f"][:N]
    order = np.argsort(F0, kind="stable")
    save(name, obj=obj, n=n, N=N, seed=seed, maxeval=me, ret=p["ret"], nevals=p["nevals"], minf=p["minf"], x=p["x"],
         words=np.uint64(p["words"]), ref_checked=ref_checked,
         init_f_every16=F0[::16].copy(), init_f_blocksum64=block_sums(F0, 64),
         init_best_rows=order[:64].astype(np.int64), init_best_f=F0[order[:64]],
         init_worst_rows=order[-256:].astype(np.int64), init_worst_f=F0[order[-256:]],
         trial_f=tr["f"][N:].copy(), trial_row=tr["row"][N:].copy(), trial_kind=tr["kind"][N:].copy(),
         trial_accepted=tr["accepted"][N:].copy(), trial_xhash=p["xhash"][N:].copy())


def crs_long_case(name, obj, n, N, extra, seed=42):
    """CRS2_LM deep into the trial loop (the steady regime of rejections and mutation blocks the benchmark times, not the
    all-accept start of a fresh population): the REAL reference and the port, bit for bit over every evaluation; stored: the
    whole trial phase (f, row, kind, accepted), the hash of x of every 16th evaluation, the best-f sequence (evaluation number
    and value of every improvement of the minimum — what bench.py compares its own stopping point with) and the running count
    of accepted trials.  bench.py --steps 20 --warmup 5 stops about 51 000 evaluations into the trial loop."""
    me = N + extra
    t0 = time.time()
    p = O.run_port_crs(obj, n, N, seed, maxeval=me, trace_cap=me + 1000, record=True)
    print(name, "port: ret", p["ret"], "nevals", p["nevals"], "minf", p["minf"], "%.0f s" % (time.time() - t0), flush=True)
    tr = p["trace"]
    assert len(tr) == p["nevals"] == len(p["fseq"]) and np.array_equal(tr["f"], p["fseq"])
    t0 = time.time()
    r = O.run_ref(19, obj, n, N, seed, maxeval=me)
    print(name, "reference: ret", r["ret"], "nevals", r["nevals"], "minf", r["minf"], "%.0f s" % (time.time() - t0), flush=True)
    assert r["ret"] == p["ret"] and r["nevals"] == p["nevals"] and r["minf"] == p["minf"]
    assert np.array_equal(r["fseq"], p["fseq"]) and np.array_equal(r["xhash"], p["xhash"]) and np.array_equal(r["x"], p["x"])
    f = tr["f"]
    run_min = np.minimum.accumulate(f)
    imp = np.flatnonzero(np.concatenate(([True], run_min[1:] < run_min[:-1])))       # 0-based evaluation index of every new minimum
    save(name, obj=obj, n=n, N=N, seed=seed, maxeval=me, ret=p["ret"], nevals=p["nevals"], minf=p["minf"], x=p["x"],
         words=np.uint64(p["words"]), ref_checked=1, init_f_every16=f[:N:16].copy(), init_f_blocksum64=block_sums(f[:N], 64),
         trial_f=f[N:].copy(), trial_row=tr["row"][N:].astype(np.int32), trial_kind=tr["kind"][N:].astype(np.int8),
         trial_accepted=tr["accepted"][N:].astype(np.int8), trial_xhash_every16=p["xhash"][N::16].copy(),
         best_eval=imp.astype(np.int64) + 1, best_f=run_min[imp].copy())


def isres_stops(ex, name, obj, n, pop, nineq, gens, seed=42):
    """ISRES stopped one evaluation into generation g + 1 (maxeval = g pop + 1), g = 1 .. gens - 1: where bench.py's ISRES leg ends
    (its hook raises force_stop when `warmup + steps` generations are done; the library sees the flag after the next generation's
    first candidate, isres.c:195-198 — the reference stops on maxeval at the same evaluation with the same minimum and the same
    stream position).  The REAL reference for every g; the port beside it for the stream position (asserted identical in what the
    reference exposes: result, count, minimum, argmin)."""
    evs = [g * pop + 1 for g in range(1, gens)]
    futs = [ex.submit(_isres_stop_one, (obj, n, pop, seed, nineq, e)) for e in evs]      # beside the caller's own long run

    def collect():
        return _isres_stops_collect(name, evs, [f.result() for f in futs])
    return collect


def _isres_stops_collect(name, evs, refs):
    for e, (r, p) in zip(evs, refs):
        print(name, "stop at", e, "reference minf", r[1], "port words", p[3], flush=True)
        assert r[0] == p[0] and r[1] == p[1] and r[2] == p[2] == e and np.array_equal(r[4], p[4])
    return dict(stop_evals=np.array(evs, dtype=np.int64), stop_minf=np.array([r[1] for r, _ in refs]),
                stop_ret=np.array([r[0] for r, _ in refs], dtype=np.int64),
                stop_words=np.array([p[3] for _, p in refs], dtype=np.uint64), stop_x=np.array([r[4] for r, _ in refs]))


def _isres_stop_one(a):
    obj, n, pop, seed, nineq, e = a
    r = O.run_ref_isres(obj, n, pop, seed, nineq, 0, maxeval=e, record=False)
    p = O.run_port_isres(obj, n, pop, seed, nineq, 0, maxeval=e, record=False)
    return (r["ret"], r["minf"], r["nevals"], 0, r["x"]), (p["ret"], p["minf"], p["nevals"], p["words"], p["x"])


def isres_case(name, obj, n, pop, nineq, gens=5, seed=42):
    """ISRES: `gens` generations' evaluations (gens - 1 full generations incl. ranking + evolve, then the last generation's
    evaluations — every candidate of generation g + 1 is a function of generation g's ranking and of the evolve step): the
    generations bench.py's ISRES leg times (1 warm-up + 3) and one more."""
    from concurrent.futures import ProcessPoolExecutor
    me = gens * pop
    ex = ProcessPoolExecutor(max_workers=4)
    stops_later = isres_stops(ex, name, obj, n, pop, nineq, gens, seed)
    t0 = time.time()
    r = O.run_ref_isres(obj, n, pop, seed, nineq, 0, maxeval=me)
    print(name, "reference: ret", r["ret"], "nevals", r["nevals"], "minf", r["minf"], "%.0f s" % (time.time() - t0), flush=True)
    t0 = time.time()
    p = O.run_port_isres(obj, n, pop, seed, nineq, 0, maxeval=me)
    print(name, "port: ret", p["ret"], "nevals", p["nevals"], "minf", p["minf"], "%.0f s" % (time.time() - t0), flush=True)
    assert r["ret"] == p["ret"] and r["nevals"] == p["nevals"] and r["minf"] == p["minf"]
    # the recording callback wraps the objective (the constraints are called directly): one record per candidate
    assert np.array_equal(r["fseq"], p["fseq"]) and np.array_equal(r["xhash"], p["xhash"]) and np.array_equal(r["x"], p["x"])
    f, pen = p["ftrace"], p["pentrace"]
    assert len(f) == me
    stops = stops_later()
    ex.shutdown()
    save(name, obj=obj, n=n, pop=pop, nineq=nineq, seed=seed, maxeval=me, ret=p["ret"], nevals=p["nevals"], minf=p["minf"],
         x=p["x"], words=np.uint64(p["words"]), ref_checked=1,
         f_every8=f[::8].copy(), pen_every8=pen[::8].copy(), f_blocksum16=block_sums(f, 16), pen_blocksum16=block_sums(pen, 16),
         f_gen2_head=f[pop:pop + 2048].copy(), pen_gen2_head=pen[pop:pop + 2048].copy(),
         f_gen_heads=np.array([f[g * pop:g * pop + 512] for g in range(gens)]), pen_gen_heads=np.array([pen[g * pop:g * pop + 512] for g in range(gens)]),
         xhash_gen2_every8=p["xhash"][pop:2 * pop:8].copy(), gens=gens, **stops)


def mlsl_case(name, obj, n, nsamp, maxeval, seed=42, local_ftol_rel=1e-8, long_maxeval=100000):
    """G_MLSL_LDS + LD_LBFGS (n > 1111: the LDS variant samples pseudo-randomly, sobolseq.c:139): the run up to `maxeval`
    evaluations: every sample's f, every local search's start, result and evaluation count, in order; the same over a longer run
    (`long_*`: iterations 1-4 complete) with the iteration boundaries, and the REAL reference stopped one evaluation into iteration
    k + 1 (maxeval = evaluations at the end of iteration k, + 1), k = 1..4: where bench.py's MLSL leg ends (its hook raises
    force_stop when `warmup + steps` iterations are done; the library sees it at the next iteration's first sample, mlsl.c:366)."""
    t0 = time.time()
    r = O.run_ref_mlsl(obj, n, nsamp, seed, alg=39, local_ftol_rel=local_ftol_rel, maxeval=maxeval)
    print(name, "reference: ret", r["ret"], "nevals", r["nevals"], "minf", r["minf"], "%.0f s" % (time.time() - t0), flush=True)
    t0 = time.time()
    p = O.run_port_mlsl(obj, n, nsamp, seed, maxeval=maxeval, local_ftol_rel=local_ftol_rel, lds=True)
    print(name, "port: ret", p["ret"], "nevals", p["nevals"], "minf", p["minf"], "iterations", p["iterations"],
          "samples", len(p["fsamp"]), "local searches", len(p["floc"]), "%.0f s" % (time.time() - t0), flush=True)
    assert r["ret"] == p["ret"] and r["nevals"] == p["nevals"] and r["minf"] == p["minf"]
    assert np.array_equal(r["fseq"], p["fseq"]) and np.array_equal(r["xhash"], p["xhash"]) and np.array_equal(r["x"], p["x"])
    # the longer run: reference == port evaluation by evaluation again, then the iteration boundaries from the port
    rl = O.run_ref_mlsl(obj, n, nsamp, seed, alg=39, local_ftol_rel=local_ftol_rel, maxeval=long_maxeval)
    pl = O.run_port_mlsl(obj, n, nsamp, seed, maxeval=long_maxeval, local_ftol_rel=local_ftol_rel, lds=True)
    assert rl["ret"] == pl["ret"] and rl["nevals"] == pl["nevals"] and rl["minf"] == pl["minf"]
    assert np.array_equal(rl["fseq"], pl["fseq"]) and np.array_equal(rl["xhash"], pl["xhash"]) and np.array_equal(rl["x"], pl["x"])
    full = int(pl["iterations"]) - 1                   # the last iteration is the one maxeval cut
    print(name, "long run: iterations", pl["iterations"], "searches by iteration", pl["it_nloc"], "evaluations", pl["it_nevals"], flush=True)
    assert full >= 4
    stops = []
    for k in range(1, 5):
        e = int(pl["it_nevals"][k - 1]) + 1
        rs = O.run_ref_mlsl(obj, n, nsamp, seed, alg=39, local_ftol_rel=local_ftol_rel, maxeval=e)
        assert rs["nevals"] == e and rs["ret"] == 5
        stops.append((e, rs["minf"], rs["x"]))
        print(name, "reference stopped at", e, "minf", rs["minf"], flush=True)
    save(name, obj=obj, n=n, nsamp=nsamp, seed=seed, maxeval=maxeval, local_ftol_rel=local_ftol_rel, ret=p["ret"],
         nevals=p["nevals"], minf=p["minf"], x=p["x"], words=np.uint64(p["words"]), ref_checked=1,
         fsamp=p["fsamp"], floc=p["floc"], eloc=p["eloc"], sloc=p["sloc"], iterations=p["iterations"],
         fseq_every4=p["fseq"][::4].copy(), fseq_blocksum16=block_sums(p["fseq"], 16),
         long_maxeval=long_maxeval, long_fsamp=pl["fsamp"], long_floc=pl["floc"], long_eloc=pl["eloc"], long_sloc=pl["sloc"],
         long_it_nloc=pl["it_nloc"][:full], long_it_nevals=pl["it_nevals"][:full], long_it_words=pl["it_words"][:full],
         stop_evals=np.array([s_[0] for s_ in stops], dtype=np.int64), stop_minf=np.array([s_[1] for s_ in stops]),
         stop_x=np.array([s_[2] for s_ in stops]))


CASES = {
    # the metric configuration (BASELINE.json "metric"; SURVEY.md §8d config 5 at pop = 1e5)
    "crs_griewank_n4096_pop1e5": lambda: crs_case("crs_griewank_n4096_pop1e5", "griewank", 4096, 100000, 500),
    # the metric configuration over the evaluations bench.py's default run times (warm-up + 20 steps of 2000 = about N + 51 000)
    "crs_griewank_n4096_pop1e5_long": lambda: crs_long_case("crs_griewank_n4096_pop1e5_long", "griewank", 4096, 100000, 60000),
    # config 2
    "crs_rastrigin_n512_pop1e5": lambda: crs_case("crs_rastrigin_n512_pop1e5", "rastrigin", 512, 100000, 5000),
    # n = 64 line of the bench
    "crs_rastrigin_n64_pop1e5": lambda: crs_case("crs_rastrigin_n64_pop1e5", "rastrigin", 64, 100000, 20000),
    # config 3
    "isres_rastrigin_n256_pop5e4_4ineq": lambda: isres_case("isres_rastrigin_n256_pop5e4_4ineq", "rastrigin", 256, 50000, 4),
    # config 4 (one GPU's view: the whole job), first iteration and the start of the second
    "mlsl_ackley_n4096_N1000": lambda: mlsl_case("mlsl_ackley_n4096_N1000", "ackley", 4096, 1000, 60000),
    # config 5: the largest population the reference's 32-bit row offsets allow at n = 4096 (N (n+1) < 2^31), port == reference ...
    "crs_griewank_n4096_pop524160": lambda: crs_case("crs_griewank_n4096_pop524160", "griewank", 4096, 524160, 300),
    # ... and the configuration itself through the port alone (the reference overflows, crs.c:101,212)
    "crs_griewank_n4096_pop1e6": lambda: crs_case("crs_griewank_n4096_pop1e6", "griewank", 4096, 1000000, 300, with_ref=False),
}

if __name__ == "__main__":
    O.build_oracle()
    names = sys.argv[1:]
    if names == ["all"]:
        names = list(CASES)
    for nm in names:
        CASES[nm]()
