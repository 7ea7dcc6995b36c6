"""Hunt for intermittent CRS2_LM divergences on the GPU (development aid; run on the MI355X box).

    python tools/stress_crs.py --seconds 240 [--tag base] [--churn 1] [--dump 1]

Every iteration runs a DRAWN configuration (objective, n, population, seed, budget — consecutive runs never share a seed or a
size, so data left behind by the previous run at the same device address is never accidentally the right data) through
libnlopt_amd.so and through the CPU oracle, and compares the full traces (tests/_crsdiag.py).  --churn interleaves small MLSL /
ISRES runs the way the test suite does (other streams, other allocation sizes).  --dump makes the engine write what its init
kernels saw (NLA_CRS_DEBUG_DIR) and, on a divergence, analyses it on the spot: stream words against the host generator, rows
against the words.  Everything found goes to gpurun_out/crs_divergence.jsonl; the summary to gpurun_out/stress_<tag>.json."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120)
ap.add_argument("--tag", default="base")
ap.add_argument("--churn", type=int, default=1)
ap.add_argument("--dump", type=int, default=0)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--golden-too", type=int, default=1)
ap.add_argument("--golden-only", type=int, default=0, help="only the golden cases (robust: far from convergence), in sorted order")
ap.add_argument("--uc-churn", type=int, default=0, help="between runs: allocate, write and free a raw uncached buffer (MB drawn up to this)")
args = ap.parse_args()

OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
DUMPDIR = os.path.join(OUT, "dump_" + args.tag)
if args.dump:
    os.makedirs(DUMPDIR, exist_ok=True)
    os.environ["NLA_CRS_DEBUG_DIR"] = DUMPDIR

import _crsdiag as D  # noqa: E402
import _oracle as O  # noqa: E402
import nlopt_amd  # noqa: E402
import test_gpu_crs as T  # noqa: E402

GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "crs_golden.json")))
OBJS = ["rastrigin", "ackley", "griewank", "rosenbrock", "levy", "sphere"]
rng = np.random.RandomState(args.seed)


def analyse_dump(seed, obj):
    """what did the init kernels see?  words vs the host generator, rows vs the words"""
    path = os.path.join(DUMPDIR, "init_last.bin")
    if not os.path.exists(path):
        return dict(dump="missing")
    raw = np.fromfile(path, dtype=np.uint8)
    hdr = raw[:40].view(np.int64)
    n, ld, N, nwords, origin = (int(v) for v in hdr)
    off = 40
    words = raw[off:off + 4 * nwords].view(np.uint32)
    off += 4 * nwords
    X = raw[off:off + 8 * N * ld].view(np.float64).reshape(N, ld)
    off += 8 * N * ld
    F = raw[off:off + 8 * N].view(np.float64)
    # the reference's stream: init_genrand(seed), first word = the `origin`-th output
    rs = np.random.RandomState(seed)          # legacy seeding == init_genrand for an integer seed
    ref = rs._bit_generator.random_raw(origin + nwords).astype(np.uint32)[origin:]
    badw = np.flatnonzero(words != ref)
    rep = dict(dump_n=n, dump_N=N, dump_origin=origin, words_bad=int(badw.size))
    if badw.size:
        runs = np.split(badw, np.flatnonzero(np.diff(badw) != 1) + 1)
        rep["words_bad_runs"] = [[int(r[0]), int(r[-1])] for r in runs[:16]]
        rep["words_bad_nruns"] = len(runs)
    lo, hi = nlopt_amd.objective_box(obj)
    a = (ref[0::2] >> 5).astype(np.float64)
    b = (ref[1::2] >> 6).astype(np.float64)
    u = (a * 67108864.0 + b) * (1.0 / 9007199254740992.0)
    Xref = lo + (hi - lo) * u.reshape(N - 1, n)
    badx = np.flatnonzero((X[1:, :n] != Xref).any(axis=1)) + 1
    rep["rows_bad"] = int(badx.size)
    if badx.size:
        rep["rows_bad_first"] = [int(v) for v in badx[:16]]
        # do the bad rows agree with the words that were in the buffer (wrong words) or not even that (wrong kernel)?
        a2 = (words[0::2] >> 5).astype(np.float64)
        b2 = (words[1::2] >> 6).astype(np.float64)
        Xw = lo + (hi - lo) * ((a2 * 67108864.0 + b2) * (1.0 / 9007199254740992.0)).reshape(N - 1, n)
        rep["rows_bad_but_consistent_with_dumped_words"] = int(np.count_nonzero((X[1:, :n] == Xw).all(axis=1)[badx - 1]))
    rep["F_head"] = [float(v).hex() for v in F[:4]]
    keep = os.path.join(DUMPDIR, "init_fail_%d.bin" % int(time.time() * 1000))
    if raw.nbytes < (8 << 20):
        os.replace(path, keep)
        rep["dump_kept"] = os.path.basename(keep)
    return rep


_gi = [0]


def draw():
    if args.golden_only:
        names = [k for k in sorted(GOLD) if GOLD[k]["nevals"] < 50000]
        name = names[_gi[0] % len(names)]
        _gi[0] += 1
        g = GOLD[name]
        return name, g["obj"], g["n"], g["pop"], g["seed"], dict(g["kwargs"]), g
    if args.golden_too and rng.rand() < 0.3:
        name = sorted(GOLD)[rng.randint(len(GOLD))]
        g = GOLD[name]
        return name, g["obj"], g["n"], g["pop"], g["seed"], dict(g["kwargs"]), g
    obj = OBJS[rng.randint(len(OBJS))]
    n = int(rng.choice([2, 3, 4, 5, 8, 10, 16, 31, 64, 100, 257, 300]))
    if obj in ("rosenbrock", "levy") and n < 2:
        n = 2
    pop = int(rng.randint(n + 2, 4 * n + 40))
    seed = int(rng.randint(1, 1 << 30))
    me = pop + int(rng.randint(200, 2500))
    return "drawn_%s_n%d_pop%d_seed%d_me%d" % (obj, n, pop, seed, me), obj, n, pop, seed, dict(maxeval=me), None


def churn(it):
    """the kind of work the suite does between CRS cases: small MLSL / ISRES runs (other streams, allocation sizes)"""
    try:
        if it % 2 == 0:
            o = nlopt_amd.Opt(nlopt_amd.G_MLSL, 4)
            lo, hi = nlopt_amd.objective_box("rastrigin")
            o.set_lower_bounds(lo); o.set_upper_bounds(hi)
            o.set_min_objective(nlopt_amd.objective("rastrigin"))
            lopt = nlopt_amd.Opt(nlopt_amd.LD_LBFGS, 4)
            lopt.set_ftol_rel(1e-8)
            nlopt_amd.lib().nlopt_set_local_optimizer(o._h, lopt._h)
            o.set_maxeval(600)
            nlopt_amd.srand(int(rng.randint(1, 1 << 30)))
            o.optimize_raw(np.full(4, 1.0))
        else:
            o = nlopt_amd.Opt(nlopt_amd.GN_ISRES, 8)
            lo, hi = nlopt_amd.objective_box("griewank")
            o.set_lower_bounds(lo); o.set_upper_bounds(hi)
            o.set_min_objective(nlopt_amd.objective("griewank"))
            o.set_population(40)
            o.set_maxeval(400)
            nlopt_amd.srand(int(rng.randint(1, 1 << 30)))
            o.optimize_raw(np.full(8, 1.0))
    except Exception as e:
        print("churn failed:", e)


PORT_CACHE = {}
t_end = time.time() + args.seconds
runs = bad = 0
kinds = {}
while time.time() < t_end:
    name, obj, n, pop, seed, kw, g = draw()
    a = T.run_amd(obj, n, pop, seed, trace_cap=200000, **kw)
    if g is not None and name in PORT_CACHE:
        p = PORT_CACHE[name]
    else:
        p = O.run_port_crs(obj, n, pop, seed, trace_cap=200000, **kw)
        if g is not None:
            PORT_CACHE[name] = p
    runs += 1
    rep = D.explain(name, a, p, g, extra=dict(tag=args.tag, iteration=runs))
    if not rep["ok"]:
        bad += 1
        kinds[rep.get("phase", "?")] = kinds.get(rep.get("phase", "?"), 0) + 1
        more = analyse_dump(seed, obj) if args.dump else {}
        print("DIVERGENCE", json.dumps(dict(rep, **more))[:3000], flush=True)
        if more:
            with open(os.path.join(OUT, "crs_divergence.jsonl"), "a") as f:
                f.write(json.dumps(dict(case=name, dump_analysis=more, tag=args.tag)) + "\n")
    if args.churn and runs % 3 == 0:
        churn(runs)
    if args.uc_churn:
        L = nlopt_amd.lib()
        L.nla_debug_uncached_churn.argtypes = [C.c_size_t]
        for _ in range(2):
            L.nla_debug_uncached_churn(int(rng.randint(1, args.uc_churn * (1 << 20))))
summary = dict(tag=args.tag, seconds=args.seconds, runs=runs, bad=bad, phases=kinds, churn=args.churn, dump=args.dump, uc_churn=args.uc_churn,
               golden_only=args.golden_only,
               env={k: v for k, v in os.environ.items() if k.startswith("NLA_")})
print("SUMMARY", json.dumps(summary))
with open(os.path.join(OUT, "stress_%s.json" % args.tag), "w") as f:
    json.dump(summary, f)
