"""One rank of a world_size-N test (launched by tests/test_multiproc.py and tests/test_gpu_multiproc.py with
RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment): joins a gloo group, builds the library's
communicator over it (host transport) and runs the named case; the result goes to <out>.rank<r>.npz.
On the GPU box every rank uses HIP device 0 (one GPU there) — the collective path is the same."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    case, out = sys.argv[1], sys.argv[2]
    args = json.loads(sys.argv[3]) if len(sys.argv) > 3 else {}
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import nlopt_amd
    if os.environ.get("NLA_TEST_EMU_DEVICE"):
        # the product's host drivers over the CPU stand-in for the device layer (oracle/emu_device.c): test-side switch only
        nlopt_amd.LIB_PATH = os.path.join(ROOT, "oracle", "libnlopt_amd_emu.so")
    import _oracle as O
    comm = nlopt_amd.Comm.from_torch_distributed()
    res = {}
    if case == "comm":
        # partition covers [0, count) exactly once; all-gather returns every rank's block in rank order
        cover = []
        for count in (1, 2, 7, 64, 1001):
            per, first, mine = comm.partition(count)
            cover.append([count, per, first, mine])
        res["cover"] = np.array(cover)
        a = (np.arange(5, dtype=np.float64) + 100.0 * rank)
        res["gathered"] = comm.allgather_host(a)
        big = np.full(300001, rank, dtype=np.uint8)
        g = comm.allgather_host(big)
        res["big_ok"] = np.array([int(all((g[r] == r).all() for r in range(world)))])
        res["counters"] = np.array([comm.counters()["collectives"], comm.counters()["bytes"]])
    elif case == "emu_crs":
        # the product's CRS driver over the CPU engine emulation, initial population produced in rank blocks
        E = O.emu()
        E.orc_emu_set_comm.argtypes = [__import__("ctypes").c_void_p]
        E.orc_emu_set_comm(comm._h)
        r = O.run_emu_crs(args["obj"], args["n"], args["pop"], args["seed"], maxeval=args["maxeval"], trace_cap=args["maxeval"] + 64)
        E.orc_emu_set_comm(None)
        res = dict(ret=np.array([r["ret"]]), minf=np.array([r["minf"]]), x=r["x"], nevals=np.array([r["nevals"]]),
                   words=np.array([r["words"]], dtype=np.uint64), f=r["trace"]["f"], row=r["trace"]["row"],
                   accepted=r["trace"]["accepted"], collectives=np.array([comm.counters()["collectives"]]))
    elif case in ("gpu_crs", "gpu_isres", "gpu_mlsl"):
        if nlopt_amd.device_count() <= 0:
            raise SystemExit("no HIP device visible")
        obj, n, seed = args["obj"], args["n"], args["seed"]
        xs, lo, hi = O.golden_x0(obj, n)
        alg = {"gpu_crs": nlopt_amd.GN_CRS2_LM, "gpu_isres": nlopt_amd.GN_ISRES, "gpu_mlsl": nlopt_amd.G_MLSL}[case]
        if case == "gpu_mlsl" and args.get("lds"):
            alg = nlopt_amd.G_MLSL_LDS
        if case == "gpu_mlsl" and args.get("local") == "default":      # GD_MLSL(_LDS): the dispatcher's default local optimiser
            alg = nlopt_amd.GD_MLSL_LDS if args.get("lds") else nlopt_amd.GD_MLSL
        o = nlopt_amd.Opt(alg, n)
        o.set_lower_bounds(lo)
        o.set_upper_bounds(hi)
        o.set_min_objective(nlopt_amd.objective(obj))
        if args.get("pop"):
            o.set_population(args["pop"])
        o.set_maxeval(args["maxeval"])
        if case == "gpu_isres" and args.get("ncon"):
            o.add_blocksum_constraints(args["ncon"], 1e-8)
        if case == "gpu_mlsl" and args.get("local") == "default":
            o.set_ftol_rel(1e-8)
        elif case == "gpu_mlsl":
            loc = nlopt_amd.Opt(nlopt_amd.LD_MMA if args.get("local") == "mma" else nlopt_amd.LD_LBFGS, n)
            loc.set_ftol_rel(1e-8)
            nlopt_amd.lib().nlopt_set_local_optimizer(o._h, loc._h)
        if args.get("sharded", True):
            o.set_comm(comm)
        o.enable_trace(args["maxeval"] + 4096)
        nlopt_amd.srand(seed)
        x, minf, ret = o.optimize_raw(xs)
        t = o.trace()
        res = dict(ret=np.array([ret]), minf=np.array([minf]), x=x, nevals=np.array([o.get_numevals()]), f=t["f"], row=t["row"],
                   kind=t["kind"], accepted=t["accepted"], collectives=np.array([comm.counters()["collectives"]]),
                   gathered_bytes=np.array([comm.counters()["bytes"]]), after=np.array([nlopt_amd.lib().nla_genrand_int32()], dtype=np.uint64))
    else:
        raise SystemExit("unknown case " + case)
    np.savez(out + ".rank%d.npz" % rank, **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
