#!/usr/bin/env python
"""bench.py — the headline benchmark of BASELINE.json on MI355X.

Metric: candidate-evals/sec (+ gens-to-ftol), NLOPT_GN_CRS2_LM, Griewank n=4096, pop=1e5
(BASELINE.json "metric"; the configuration fits one GPU: 3.3 GB of 288 GB).

A *step* is one pass of the hot path over one batch of work: `--evals-per-step` candidate
evaluations of the CRS2_LM trial loop (reflection gather-sum + objective + in-order commit,
src/algs/crs/crs.c:125-156) on a population that is already resident in HBM.  The population
initialisation (crs_init) happens before the timed region and is reported separately
(`init_evals_per_s`).  W warm-up steps, then EXACTLY K timed steps bracketed by a barrier and a
device synchronisation; value = evaluations made in the K steps / wall time, max over ranks.

N > 1 (launched by torchrun, one rank per GPU): the CRS2_LM trial loop is one serial accept/reject
chain — in this round each rank runs an independent replica of the workload on its
own GPU with its own seed ("replicas only", scaling = weak, no data-path collective); `value` is
the sum over ranks.  See DESIGN.md §multi-GPU for what comes next.

Output: ONE JSON line on rank 0 with the driver's contract fields plus
  roofline     — dominant kernel (crs_advance_kernel, the resumable gather-sum): algorithmic bytes
                 = 8n per population row summed (n+1 rows = 8 n (n+1) B per trial), counted per
                 launch from the slots' progress / HIP-event time of those launches on their own
                 stream, vs 8 TB/s HBM
  cpu_baseline — the real reference NLopt (oracle/_ref, kind "reference") or the C port (kind
                 "port") timed single-threaded on this host on a bounded sample of the same workload
  gens_to_ftol — numevals/pop at NLOPT_FTOL_REACHED on the small configuration where the CPU
                 reference can reach it, with the reference's golden value beside it.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--pop", type=int, default=100000)
    ap.add_argument("--obj", default="griewank")
    ap.add_argument("--evals-per-step", type=int, default=2000)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--max-spec", type=int, default=0)
    ap.add_argument("--gather-variant", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-pop", type=int, default=20000)
    ap.add_argument("--cpu-sample-trials", type=int, default=400)
    return ap.parse_args()


def cpu_baseline(obj, n, pop_sample, trials, seed):
    """single-thread CPU reference on a bounded sample: init (untimed) + `trials` trial-phase
    evaluations of the same workload at the same n (the trial-phase rate does not depend on pop:
    SURVEY.md §6, 49.9 vs 50.3 evals/s at pop 2e4 vs 1e5)."""
    import _oracle as O

    class Timer(C.Structure):
        _fields_ = [("inner", C.c_void_p), ("inner_data", C.c_void_p), ("mark", C.c_long), ("count", C.c_long),
                    ("t_first", C.c_double), ("t_mark", C.c_double), ("t_last", C.c_double)]

    P = O.port()
    xs, lo, hi = O.golden_x0(obj, n)
    x = np.array(xs)
    lb, ub = np.full(n, lo), np.full(n, hi)
    tm = Timer(P.orc_objective(O.OBJ[obj]), None, pop_sample, 0, 0.0, 0.0, 0.0)
    cb = C.cast(P.orc_timing_callback, C.c_void_p).value
    me = pop_sample + trials
    minf = C.c_double()
    if O.have_ref():
        R = O.ref()
        opt = R.nlopt_create(19, n)
        R.nlopt_set_lower_bounds(opt, O.dptr(lb))
        R.nlopt_set_upper_bounds(opt, O.dptr(ub))
        R.nlopt_set_min_objective(opt, cb, C.cast(C.pointer(tm), C.c_void_p))
        R.nlopt_set_population(opt, pop_sample)
        R.nlopt_set_maxeval(opt, me)
        R.nlopt_srand(seed)
        R.nlopt_optimize(opt, O.dptr(x), C.byref(minf))
        R.nlopt_destroy(opt)
        kind = "reference"
    else:
        st = O.OrcStop()
        P.orc_stop_default(C.byref(st), n)
        st.maxeval = me
        P.orc_srand(seed)
        P.orc_crs_minimize(n, cb, C.cast(C.pointer(tm), C.c_void_p), O.dptr(lb), O.dptr(ub), O.dptr(x), C.byref(minf),
                           C.byref(st), pop_sample, None)
        kind = "port"
    ntrial = tm.count - tm.mark
    dt = tm.t_last - tm.t_mark
    return dict(value=ntrial / dt if dt > 0 else None, unit="evals/s", cores=1, kind=kind,
                sample="NLOPT_GN_CRS2_LM %s n=%d, pop=%d init untimed (%.1f s, %.0f evals/s), then %d trial-phase evals in %.1f s, 1 thread"
                       % (obj, n, pop_sample, tm.t_mark - tm.t_first, tm.mark / max(tm.t_mark - tm.t_first, 1e-9), ntrial, dt),
                init_evals_per_s=tm.mark / max(tm.t_mark - tm.t_first, 1e-9))


def gens_to_ftol():
    """gens-to-ftol on the configuration where the reference can reach it:
    CRS2_LM Rastrigin n=10 pop=100 ftol_rel=1e-4, seed 42 — golden: 5385 evals = 53.85 'generations'."""
    import nlopt_amd
    import _oracle as O
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "crs_golden.json")))["rastrigin_n10_pop100_ftol1e-4"]
    xs, lo, hi = O.golden_x0("rastrigin", 10)
    o = nlopt_amd.Opt(nlopt_amd.GN_CRS2_LM, 10)
    o.set_lower_bounds(lo)
    o.set_upper_bounds(hi)
    o.set_min_objective(nlopt_amd.objective("rastrigin"))
    o.set_population(100)
    o.set_ftol_rel(1e-4)
    nlopt_amd.srand(42)
    x, minf, ret = o.optimize_raw(xs)
    return dict(config="NLOPT_GN_CRS2_LM rastrigin n=10 pop=100 ftol_rel=1e-4 seed=42", result=int(ret), numevals=o.get_numevals(),
                value=o.get_numevals() / 100.0, minf=minf, reference_numevals=gold["nevals"],
                reference_value=gold["nevals"] / 100.0, reference_minf=float.fromhex(gold["minf"]))


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    import nlopt_amd
    L = nlopt_amd.lib()
    if nlopt_amd.device_count() <= 0:
        raise SystemExit("bench.py: no HIP device visible (libnlopt_amd has no CPU fallback)")
    L.nla_dev_set.argtypes = [C.c_int]
    L.nla_dev_set(local_rank)
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)

    import _oracle as O
    n, pop = a.n, a.pop
    xs, lo, hi = O.golden_x0(a.obj, n)
    o = nlopt_amd.Opt(nlopt_amd.GN_CRS2_LM, n)
    o.set_lower_bounds(lo)
    o.set_upper_bounds(hi)
    o.set_min_objective(nlopt_amd.objective(a.obj))
    o.set_population(pop)
    if a.max_spec:
        o.set_param("amd_max_spec", a.max_spec)
    if a.gather_variant:
        o.set_param("amd_gather_variant", a.gather_variant)
    nlopt_amd.srand(a.seed + rank)
    x = np.array(xs)
    minf, ret = C.c_double(), C.c_int()
    t0 = time.perf_counter()
    s = L.nlopt_amd_crs_open(o._h, x.ctypes.data_as(C.POINTER(C.c_double)), C.byref(minf), C.byref(ret))
    t_init = time.perf_counter() - t0
    if not s or ret.value != 1:
        raise SystemExit("bench.py: crs_open failed: ret=%d %s" % (ret.value, o.get_errmsg()))

    def sync_all():
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        L.nla_stream_sync(None)
        if dist is not None:
            dist.barrier()

    for _ in range(a.warmup):
        if L.nlopt_amd_crs_step(s, a.evals_per_step) != 1:
            raise SystemExit("bench.py: the run stopped during warm-up")
    sync_all()
    st0, ev0 = o.stats(), o.get_numevals()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        if L.nlopt_amd_crs_step(s, a.evals_per_step) != 1:
            raise SystemExit("bench.py: the run stopped inside the timed region")
    sync_all()
    dt = time.perf_counter() - t0
    st1, ev1 = o.stats(), o.get_numevals()
    fret = L.nlopt_amd_crs_close(s)
    evals = ev1 - ev0

    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt_max = float(tt.item())
        te = torch.tensor([float(evals)], dtype=torch.float64, device="cuda")
        dist.all_reduce(te, op=dist.ReduceOp.SUM)
        evals_all = float(te.item())
    else:
        dt_max, evals_all = dt, float(evals)

    if rank == 0:
        g_ms = st1["t_gather_ms"] - st0["t_gather_ms"]
        g_bytes = st1["gather_bytes"] - st0["gather_bytes"]
        g_launch = st1["gather_launches"] - st0["gather_launches"]
        achieved = (g_bytes / 1e9) / (g_ms / 1e3) if g_ms > 0 else None
        slots = st1["slots_launched"] - st0["slots_launched"]
        used = st1["slots_used"] - st0["slots_used"]
        out = {
            "metric": "candidate-evals/sec, CRS2_LM n=%d pop=%d (trial phase)" % (n, pop),
            "value": evals_all / dt_max, "unit": "evals/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt_max / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "NLOPT_GN_CRS2_LM %s n=%d pop=%d seed=%d, %d candidate evals per step, population resident in HBM%s"
                                   % (a.obj, n, pop, a.seed, a.evals_per_step,
                                      "" if world == 1 else "; %d independent replicas (seed+rank)" % world),
                       "evals_timed": int(evals_all), "evals_per_step_requested": a.evals_per_step},
            "roofline": {"bound": "hbm", "kernel": "crs_advance_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": None,
                         "launches": int(g_launch), "avg_launch_ms": (g_ms / g_launch) if g_launch else None,
                         "algorithmic_bytes_per_trial": 8 * n * (n + 1),
                         "avg_algorithmic_MB_per_launch": (g_bytes / 1e6 / g_launch) if g_launch else None,
                         "avg_trials_consumed_per_launch": (used / g_launch) if g_launch else None},
            "window": {"slots_started": int(slots), "slots_used": int(used),
                            "useful_frac": (used / slots) if slots else None,
                            "newbest": int(st1["slots_newbest"] - st0["slots_newbest"]),
                            "role": int(st1["slots_role"] - st0["slots_role"]),
                            "accepted": int(st1["accepted"] - st0["accepted"])},
            "init": {"evals": pop, "seconds": t_init, "init_evals_per_s": pop / t_init},
            "final_result": int(fret), "minf": minf.value,
        }
        try:
            out["gens_to_ftol"] = gens_to_ftol()
        except Exception as e:        # the headline line must still be printed
            out["gens_to_ftol"] = {"error": repr(e)}
        if not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(a.obj, n, a.cpu_sample_pop, a.cpu_sample_trials, a.seed)
                if out["cpu_baseline"]["value"]:
                    out["speedup_vs_cpu_single_thread"] = out["value"] / out["cpu_baseline"]["value"]
            except Exception as e:
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
