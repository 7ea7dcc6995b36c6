"""Generate tests/golden/crs_golden.json, isres_golden.json and rest_golden.json (LD_LBFGS, LD_MMA, G_MLSL / GD_MLSL incl. Sobol sampling, GN_ESCH,
the Sobol sequence) from the REAL reference (oracle/_ref/libnlopt_ref.so, built
from /root/reference by oracle/Makefile).  Run in the build container only:
    python tests/golden/make_golden.py
Each case records what a later run (port or HIP path) must reproduce: nlopt_result, numevals, minf
(hex, bit-exact for the CPU port; 1e-10 relative for the device objective), the argmin x (hex),
and the full per-evaluation f sequence compressed to a hash + the first/last few values."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _oracle as O  # noqa: E402

CASES = [
    # (name, obj, n, pop, seed, kwargs)  — kwargs go to both the reference and the candidates
    ("cfg1_rosenbrock_n10_pop100", "rosenbrock", 10, 100, 42, dict(maxeval=10000)),
    ("rastrigin_n10_pop100_ftol1e-4", "rastrigin", 10, 100, 42, dict(ftol_rel=1e-4)),
    ("rastrigin_n10_pop100_ftol1e-8", "rastrigin", 10, 100, 42, dict(ftol_rel=1e-8)),
    ("griewank_n8_pop50", "griewank", 8, 50, 42, dict(maxeval=5002)),
    ("griewank_n10_default_pop", "griewank", 10, 0, 12345, dict(maxeval=3000)),
    ("levy_n4_default_pop", "levy", 4, 0, 12345, dict(maxeval=3000)),
    ("ackley_n16_pop400", "ackley", 16, 400, 7, dict(maxeval=6000)),
    ("rastrigin_n64_pop2000", "rastrigin", 64, 2000, 42, dict(maxeval=12000)),
    ("sphere_n3_pop4_stopval", "sphere", 3, 4, 3, dict(stopval=1e-3, maxeval=4000)),
    ("sphere_n3_pop40_stopval_hit", "sphere", 3, 40, 3, dict(stopval=1e-2, maxeval=4000)),
    ("rastrigin_n10_stopval_in_init", "rastrigin", 10, 100, 42, dict(stopval=120.0)),
    ("rastrigin_n10_maxeval_in_init", "rastrigin", 10, 100, 42, dict(maxeval=37)),
    ("rosenbrock_n5_xtol", "rosenbrock", 5, 60, 9, dict(xtol_rel=1e-3, maxeval=20000)),
    ("rastrigin_n257_pop600_odd_n", "rastrigin", 257, 600, 5, dict(maxeval=1500)),
    ("griewank_n512_pop3000", "griewank", 512, 3000, 42, dict(maxeval=5000)),
    # BASELINE.md's second gens-to-ftol pin: 191387 evaluations = 95.69 generations, minf 7.9706244871590783
    ("rastrigin_n64_pop2000_ftol1e-6", "rastrigin", 64, 2000, 42, dict(ftol_rel=1e-6)),
]


def fhash(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float64).tobytes()).hexdigest()


ISRES_CASES = [
    # (name, obj, n, pop, seed, nineq, neq, kwargs)
    ("isres_rastrigin_n8_pop40_uncon", "rastrigin", 8, 40, 42, 0, 0, dict(maxeval=2000)),
    ("isres_rastrigin_n12_pop60_4ineq", "rastrigin", 12, 60, 42, 4, 0, dict(maxeval=3000)),
    ("isres_griewank_n6_defaultpop_2ineq_1eq", "griewank", 6, 0, 7, 2, 1, dict(maxeval=2500)),
    ("isres_sphere_n5_pop30_ftolabs", "sphere", 5, 30, 3, 1, 0, dict(ftol_abs=1e-9, maxeval=20000)),
    ("isres_rosenbrock_n4_pop35_xtol", "rosenbrock", 4, 35, 11, 0, 0, dict(xtol_rel=1e-4, maxeval=20000)),
    ("isres_levy_n3_pop7_2eq", "levy", 3, 7, 5, 0, 2, dict(maxeval=700)),
    ("isres_sphere_n4_pop50_stopval", "sphere", 4, 50, 9, 2, 0, dict(stopval=0.05, maxeval=20000)),
    ("isres_rastrigin_n32_pop700_4ineq", "rastrigin", 32, 700, 42, 4, 0, dict(maxeval=7000)),
    ("isres_ackley_n64_pop300_uncon", "ackley", 64, 300, 5, 0, 0, dict(maxeval=3000)),
    ("isres_rastrigin_n10_maxeval_midgen", "rastrigin", 10, 100, 42, 2, 0, dict(maxeval=257)),
]


def main_isres():
    out = {}
    for name, obj, n, pop, seed, nineq, neq, kw in ISRES_CASES:
        r = O.run_ref_isres(obj, n, pop, seed, nineq, neq, **kw)
        out[name] = dict(obj=obj, n=n, pop=pop, seed=seed, nineq=nineq, neq=neq, kwargs=kw, ret=int(r["ret"]),
                         nevals=int(r["nevals"]), minf=float(r["minf"]).hex(), x=[float(v).hex() for v in r["x"]],
                         fseq_sha256=fhash(r["fseq"]), xhash_sha256=hashlib.sha256(r["xhash"].tobytes()).hexdigest(),
                         fseq_head=[float(v).hex() for v in r["fseq"][:8]], fseq_tail=[float(v).hex() for v in r["fseq"][-8:]])
        print(name, r["ret"], r["nevals"], r["minf"])
    with open(os.path.join(HERE, "isres_golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


CHK = 50     # evaluations per checkpoint of the decision sequence


def decision_checkpoints(trace, fseq):
    """The run as checkpoints a test can localise a divergence with on ANY machine: per block of CHK evaluations the CRC32 of
    its (row, kind, accepted) records, and the f of the block's last evaluation (hex; compared to 1e-10 by the tests)."""
    import zlib
    rec = np.zeros(len(trace), dtype=[("row", "<i8"), ("kind", "<i4"), ("accepted", "<i4")])
    rec["row"], rec["kind"], rec["accepted"] = trace["row"], trace["kind"], trace["accepted"]
    crc, fl = [], []
    for i in range(0, len(rec), CHK):
        crc.append(zlib.crc32(rec[i:i + CHK].tobytes()) & 0xffffffff)
        fl.append(float(fseq[min(i + CHK, len(rec)) - 1]).hex())
    return dict(every=CHK, crc32=crc, f_last=fl)


def main():
    out = {}
    for name, obj, n, pop, seed, kw in CASES:
        r = O.run_ref(19, obj, n, pop, seed, cap=400000, **kw)
        # which row every evaluation went to is not visible through the reference's API: it comes from the 64-bit port, whose
        # f sequence must be the reference's own, evaluation by evaluation and bit for bit, for it to count
        p = O.run_port_crs(obj, n, pop, seed, trace_cap=len(r["fseq"]) + 8, **kw)
        assert p["ret"] == r["ret"] and p["nevals"] == r["nevals"] and len(p["trace"]) == len(r["fseq"]), name
        assert np.array_equal(p["trace"]["f"], r["fseq"]) and np.array_equal(p["x"], r["x"]), name
        out[name] = dict(obj=obj, n=n, pop=pop, seed=seed, kwargs=kw, ret=int(r["ret"]), nevals=int(r["nevals"]),
                         minf=float(r["minf"]).hex(), x=[float(v).hex() for v in r["x"]],
                         fseq_sha256=fhash(r["fseq"]), xhash_sha256=hashlib.sha256(r["xhash"].tobytes()).hexdigest(),
                         fseq_head=[float(v).hex() for v in r["fseq"][:8]], fseq_tail=[float(v).hex() for v in r["fseq"][-8:]],
                         checkpoints=decision_checkpoints(p["trace"], r["fseq"]))
        print(name, r["ret"], r["nevals"], r["minf"])
    with open(os.path.join(HERE, "crs_golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


def record(r):
    return dict(ret=int(r["ret"]), nevals=int(r["nevals"]), minf=float(r["minf"]).hex(), x=[float(v).hex() for v in r["x"]],
                fseq_sha256=fhash(r["fseq"]), xhash_sha256=hashlib.sha256(r["xhash"].tobytes()).hexdigest(),
                fseq_head=[float(v).hex() for v in r["fseq"][:8]], fseq_tail=[float(v).hex() for v in r["fseq"][-8:]])


# the rest of the path: LD_LBFGS, G_MLSL (pseudo-random and Sobol sampling), GN_ESCH, and the Sobol sequence itself
LBFGS_CASES = [
    ("lbfgs_rosenbrock_n10", "rosenbrock", 10, dict(ftol_rel=1e-10, maxeval=2000)),
    ("lbfgs_ackley_n64", "ackley", 64, dict(ftol_rel=1e-8, maxeval=500)),
    ("lbfgs_rastrigin_n30_xtol", "rastrigin", 30, dict(xtol_rel=1e-8, maxeval=500)),
    ("lbfgs_griewank_n200_mf5", "griewank", 200, dict(ftol_rel=1e-9, maxeval=400, mf=5)),
]
MLSL_CASES = [
    ("mlsl_rastrigin_n4_ns10", "rastrigin", 4, 10, 42, 38, dict(maxeval=3000)),
    ("mlsl_ackley_n6_ns25", "ackley", 6, 25, 7, 38, dict(maxeval=5000)),
    ("mlsl_levy_n3_default", "levy", 3, 0, 11, 38, dict(maxeval=1500)),
    ("mlsl_lds_rastrigin_n4_ns10_sobol", "rastrigin", 4, 10, 42, 39, dict(maxeval=3000)),
    ("mlsl_lds_griewank_n30_ns50_sobol", "griewank", 30, 50, 1, 39, dict(maxeval=6000)),
    ("mlsl_lds_ackley_n1200_pseudo", "ackley", 1200, 30, 4, 39, dict(maxeval=2600, local_ftol_rel=1e-6)),   # n > 1111: no Sobol generator
]
ESCH_CASES = [
    ("esch_rastrigin_n6_default", "rastrigin", 6, 0, 42, dict(maxeval=3000)),
    ("esch_griewank_n10_pop50", "griewank", 10, 50, 7, dict(maxeval=4000)),
    ("esch_sphere_n1_pop5", "sphere", 1, 5, 5, dict(maxeval=400)),
    ("esch_rosenbrock_n30_pop200", "rosenbrock", 30, 200, 11, dict(maxeval=6000)),
    ("esch_levy_n8_stopval", "levy", 8, 30, 1, dict(stopval=0.5, maxeval=20000)),
]
MMA_CASES = [     # NLOPT_LD_MMA without nonlinear constraints (the GD_MLSL default local optimiser)
    ("mma_rosenbrock_n10", "rosenbrock", 10, dict(maxeval=500)),
    ("mma_ackley_n64", "ackley", 64, dict(ftol_rel=1e-8)),
    ("mma_griewank_n12_xtol", "griewank", 12, dict(xtol_rel=1e-6)),
    ("mma_rastrigin_n16_nograd", "rastrigin", 16, dict(ftol_rel=1e-9, params=dict(inner_gradients=0))),
    ("mma_levy_n7_original_rule", "levy", 7, dict(ftol_abs=1e-12, params=dict(always_improve=0, rho_init=0.1))),
    ("mma_rastrigin_n12_step", "rastrigin", 12, dict(ftol_rel=1e-9, step=0.3)),
    ("mma_ackley_n6_nograd_innermax", "ackley", 6, dict(maxeval=300, params=dict(inner_gradients=0, inner_maxeval=1, always_improve=0))),
]
MLSL_MMA_CASES = [   # (name, obj, n, ns, seed, alg, local, kw): local None = the dispatcher's default local optimiser
    ("gd_mlsl_default_rastrigin_n6", "rastrigin", 6, 20, 11, 21, None, dict(maxeval=3000, ftol_rel=1e-7)),
    ("gd_mlsl_lds_default_ackley_n10", "ackley", 10, 0, 11, 23, None, dict(maxeval=3000, ftol_rel=1e-7)),
    ("g_mlsl_mma_griewank_n8", "griewank", 8, 0, 7, 38, "mma", dict(maxeval=2500)),
    ("g_mlsl_mma_nograd_rastrigin_n5", "rastrigin", 5, 12, 5, 38, "mma", dict(maxeval=1500, local_params=dict(inner_gradients=0))),
]
SOBOL_CASES = [(1, 0, 64), (2, 0, 64), (7, 110, 32), (40, 1000, 16), (1111, 11114, 4)]


def main_rest():
    out = dict(lbfgs={}, mlsl={}, esch={}, sobol={}, mma={}, mlsl_mma={})
    for name, obj, n, kw in MMA_CASES:
        r = O.run_ref_mma(obj, n, **kw)
        out["mma"][name] = dict(obj=obj, n=n, kwargs=kw, **record(r))
        print(name, r["ret"], r["nevals"], r["minf"])
    for name, obj, n, ns, seed, alg, local, kw in MLSL_MMA_CASES:
        r = O.run_ref_mlsl(obj, n, ns, seed, alg=alg, local=local, **kw)
        out["mlsl_mma"][name] = dict(obj=obj, n=n, ns=ns, seed=seed, alg=alg, local=local, kwargs=kw, **record(r))
        print(name, r["ret"], r["nevals"], r["minf"])
    for name, obj, n, kw in LBFGS_CASES:
        r = O.run_ref_lbfgs(obj, n, **kw)
        out["lbfgs"][name] = dict(obj=obj, n=n, kwargs=kw, **record(r))
        print(name, r["ret"], r["nevals"], r["minf"])
    for name, obj, n, ns, seed, alg, kw in MLSL_CASES:
        r = O.run_ref_mlsl(obj, n, ns, seed, alg=alg, **kw)
        out["mlsl"][name] = dict(obj=obj, n=n, ns=ns, seed=seed, alg=alg, kwargs=kw, **record(r))
        print(name, r["ret"], r["nevals"], r["minf"])
    for name, obj, n, pop, seed, kw in ESCH_CASES:
        r = O.run_ref_esch(obj, n, pop, seed, **kw)
        out["esch"][name] = dict(obj=obj, n=n, pop=pop, seed=seed, kwargs=kw, **record(r))
        print(name, r["ret"], r["nevals"], r["minf"])
    for sdim, skip_n, count in SOBOL_CASES:
        pts = O.ref_sobol_points(sdim, skip_n, count)
        out["sobol"]["sdim%d_skip%d" % (sdim, skip_n)] = dict(sdim=sdim, skip_n=skip_n, count=count, sha256=fhash(pts),
                                                                first=[float(v).hex() for v in pts[0][:8]],
                                                                last=[float(v).hex() for v in pts[-1][:8]])
    with open(os.path.join(HERE, "rest_golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    if "rest" in sys.argv[1:] or len(sys.argv) == 1:
        main_rest()
    if "isres" in sys.argv[1:] or len(sys.argv) == 1:
        main_isres()
    if "crs" in sys.argv[1:] or len(sys.argv) == 1:
        main()
