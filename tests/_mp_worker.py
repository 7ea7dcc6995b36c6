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
    if os.environ.get("NLA_TEST_MOCK_RCCL"):
        # the library's RCCL transport (comm.c: ncclCommInitRank / ncclAllGather on "device" buffers) bound to oracle/libmockrccl.so, which
        # moves the data through shared memory and checks the collective contract; the unique id travels over the gloo group as the
        # launcher of a real run would hand it on
        os.environ["NLA_RCCL_LIBRARY"] = os.path.join(ROOT, "oracle", "libmockrccl.so")
        uid = [nlopt_amd.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        comm = nlopt_amd.Comm.rccl(rank, world, uid[0])
    elif os.environ.get("NLA_TEST_SHM"):
        # the library's own shared-memory transport (comm.c): no Python in the exchange; the gloo group only starts the ranks together
        comm = nlopt_amd.Comm.shm(rank, world, "/nla_test_%s" % os.environ["MASTER_PORT"], int(os.environ.get("NLA_TEST_SHM_SLOT", "0")))
    else:
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
        if args.get("fix_last"):                     # a fixed coordinate: the elimination wrapper puts a host function in front
            lo_v, hi_v = np.full(n, lo), np.full(n, hi)
            lo_v[n - 1] = hi_v[n - 1] = 0.5 * (lo + hi) + 0.1
            o.set_lower_bounds(lo_v)
            o.set_upper_bounds(hi_v)
            xs = list(np.clip(np.array(xs, dtype=float), lo_v, hi_v))
        if args.get("maximize"):
            o.set_max_objective(nlopt_amd.objective(obj))
        else:
            o.set_min_objective(nlopt_amd.objective(obj))
        if args.get("pop"):
            o.set_population(args["pop"])
        o.set_maxeval(args["maxeval"])
        if args.get("stopval") is not None:
            o.set_stopval(args["stopval"])
        if args.get("maxtime_rank") is not None and rank == args["maxtime_rank"]:
            o.set_maxtime(args["maxtime"])            # ONE rank's clock runs out: all ranks must leave together
        if args.get("force_stop_rank") is not None and rank == args["force_stop_rank"]:
            import threading
            threading.Timer(args["force_stop_after"], o.force_stop).start()   # ONE rank's user raises force_stop
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
        for k, v in (args.get("params") or {}).items():
            o.set_param(k, v)
        if args.get("param_by_rank"):
            o.set_param("amd_window_factor", 2.0 + rank)          # the user's mistake of configuring the ranks differently
        if args.get("ftol_rel"):
            o.set_ftol_rel(args["ftol_rel"])
        if args.get("xtol_rel"):
            o.set_xtol_rel(args["xtol_rel"])
        o.enable_trace(args["maxeval"] + 4096)
        nlopt_amd.srand(seed + (rank if args.get("seed_by_rank") else 0))     # seed_by_rank: the user's mistake of seeding the ranks differently
        if args.get("x0_by_rank"):
            xs = list(np.array(xs, dtype=float) + 1e-3 * rank)
        x, minf, ret = o.optimize_raw(xs)
        if args.get("want_errmsg"):
            res_msg = o.get_errmsg() or ""
        if args.get("twice"):                        # the same object again, generator continuing: a second, different run
            first_run = (x.copy(), minf, ret, o.get_numevals())
            x, minf, ret = o.optimize_raw(xs)
            res_first = dict(x1=first_run[0], minf1=np.array([first_run[1]]), ret1=np.array([first_run[2]]), nevals1=np.array([first_run[3]]))
        t = o.trace()
        res = dict(ret=np.array([ret]), minf=np.array([minf]), x=x, nevals=np.array([o.get_numevals()]), f=t["f"], row=t["row"],
                   kind=t["kind"], accepted=t["accepted"], collectives=np.array([comm.counters()["collectives"]]),
                   gathered_bytes=np.array([comm.counters()["bytes"]]), after=np.array([nlopt_amd.lib().nla_genrand_int32()], dtype=np.uint64),
                   stats_allgather_bytes=np.array([o.stats()["allgather_bytes"]], dtype=np.uint64), rounds=np.array([o.stats()["rounds"]]))
        if args.get("twice"):
            res.update(res_first)
        if args.get("want_errmsg"):
            res["errmsg"] = np.array(res_msg)
    elif case == "fault_setup":
        # one rank's set-up fails (an allocation of the run's set-up phase, each in turn): every rank must come back with an error — none
        # may be left waiting in a collective — and the communicator must still serve the next, clean run
        import ctypes as C
        import torch
        L = nlopt_amd.lib()
        L.orc_emu_fail_alloc_at.argtypes = [C.c_long]
        L.orc_emu_allocs.restype = C.c_long
        seen = []                                    # (payload bytes, allocations so far) of every collective of the current run

        def allgather(b):
            seen.append((len(b), L.orc_emu_allocs()))
            t = torch.frombuffer(bytearray(b), dtype=torch.uint8)
            out = torch.empty(world * t.numel(), dtype=torch.uint8)
            dist.all_gather_into_tensor(out, t)
            return out.numpy().tobytes()
        comm = nlopt_amd.Comm.host(rank, world, allgather)
        alg, n, fail_rank = args["alg"], args["n"], args["fail_rank"]
        xs, lo, hi = O.golden_x0("rastrigin", n)

        def run(fail_at):
            a = {"crs": nlopt_amd.GN_CRS2_LM, "crs_replicas": nlopt_amd.GN_CRS2_LM, "isres": nlopt_amd.GN_ISRES, "mlsl": nlopt_amd.G_MLSL_LDS}[alg]
            o = nlopt_amd.Opt(a, n)
            o.set_lower_bounds(lo); o.set_upper_bounds(hi); o.set_min_objective(nlopt_amd.objective("rastrigin"))
            o.set_maxeval(args["maxeval"]); o.set_population(args["pop"])
            if alg == "crs_replicas":
                o.set_param("amd_shard", 0)
            if alg == "isres":
                o.add_blocksum_constraints(2, 1e-8)
            if alg == "mlsl":
                loc = nlopt_amd.Opt(nlopt_amd.LD_LBFGS, n)
                loc.set_ftol_rel(1e-6)
                L.nlopt_set_local_optimizer(o._h, loc._h)
            o.set_comm(comm)
            nlopt_amd.srand(args["seed"])
            del seen[:]
            L.orc_emu_fail_alloc_at(fail_at if rank == fail_rank else 0)
            x, minf, ret = o.optimize_raw(xs)
            L.orc_emu_fail_alloc_at(0)
            return ret, minf, np.array(x), o.get_numevals(), o.get_errmsg() or ""
        import gc
        L.orc_emu_live.restype = C.c_long
        base = run(0)
        run(0)                                       # (again: the communicator's staging was allocated by the first run and stays — the count below is that of every later run)
        gc.collect()
        live0 = L.orc_emu_live()                     # (the communicator's own staging buffers live as long as it does)
        # the allocations of the failing rank's set-up = those made before the last "ready" exchange (16-byte payload, comm.c)
        ready = [al for (nb, al) in seen if nb == 16]
        nset = [ready[-1] if ready else 0]
        dist.broadcast_object_list(nset, src=fail_rank)
        rets, msgs = [], []
        for k in range(1, nset[0] + 1):
            r = run(k)
            rets.append(r[0]); msgs.append(r[4])
        again = run(0)
        gc.collect()
        res_live = L.orc_emu_live() - live0          # device-layer objects the failed set-ups left behind
        res = dict(live=np.array([res_live]), base_ret=np.array([base[0]]), base_minf=np.array([base[1]]), base_x=base[2], base_nevals=np.array([base[3]]),
                   nset=np.array(nset), nready=np.array([len(ready)]), rets=np.array(rets), msgs=np.array(msgs),
                   again_ret=np.array([again[0]]), again_minf=np.array([again[1]]), again_x=again[2], again_nevals=np.array([again[3]]))
    elif case == "gpu_crs_rate":
        # tools/shard_probe.py: the trial-phase rate of one CRS2_LM job (population initialisation untimed), as bench.py measures it
        import ctypes as C
        import time
        L = nlopt_amd.lib()
        obj, n, pop, seed = args["obj"], args["n"], args["pop"], args["seed"]
        xs, lo, hi = O.golden_x0(obj, n)
        o = nlopt_amd.Opt(nlopt_amd.GN_CRS2_LM, n)
        o.set_lower_bounds(lo); o.set_upper_bounds(hi); o.set_min_objective(nlopt_amd.objective(obj)); o.set_population(pop)
        for k, v in (args.get("params") or {}).items():
            o.set_param(k, v)
        if world > 1:
            o.set_comm(comm)
        nlopt_amd.srand(seed)
        x = np.array(xs)
        minf, ret = C.c_double(), C.c_int()
        dist.barrier()
        t0 = time.perf_counter()
        s = L.nlopt_amd_crs_open(o._h, x.ctypes.data_as(C.POINTER(C.c_double)), C.byref(minf), C.byref(ret))
        t_init = time.perf_counter() - t0
        assert s and ret.value == 1, (ret.value, o.get_errmsg())
        r1 = L.nlopt_amd_crs_step(s, args["evals"] // 4)
        dist.barrier()
        st0, ev0, t0 = o.stats(), o.get_numevals(), time.perf_counter()
        r2 = L.nlopt_amd_crs_step(s, args["evals"])
        dist.barrier()
        dt = time.perf_counter() - t0
        st1, ev1 = o.stats(), o.get_numevals()
        L.nlopt_amd_crs_close(s)
        res = dict(evals_per_s=np.array([(ev1 - ev0) / dt]), passes=np.array([st1["rounds"] - st0["rounds"]]), dt=np.array([dt]), t_init=np.array([t_init]),
                   evals=np.array([ev1 - ev0]), gather_ms=np.array([st1["t_gather_ms"] - st0["t_gather_ms"]]),
                   gather_launches=np.array([st1["gather_launches"] - st0["gather_launches"]]),
                   allgather_bytes=np.array([st1["allgather_bytes"] - st0["allgather_bytes"]], dtype=np.uint64), minf=np.array([minf.value]),
                   step_ret=np.array([r1, r2]), errmsg=np.array(o.get_errmsg() or ""))
    elif case == "emu_sweep":
        # drawn configurations of the ISRES and MLSL host drivers over the emulated device, each checked against the oracle here
        # (every rank runs the same draws; the multi-rank runs shard them)
        L = nlopt_amd.lib()
        checked = 0
        for draw in range(args["first"], args["first"] + args["count"]):
            rng = np.random.default_rng(4242 + draw)
            obj = ["rastrigin", "ackley", "griewank", "rosenbrock", "levy", "sphere"][int(rng.integers(6))]
            n = int(rng.integers(2, 16))
            seed = int(rng.integers(1, 2 ** 31))
            xs, lo, hi = O.golden_x0(obj, n)
            # CRS2_LM: the whole product path (crs_driver.c over crs_engine.c over the emulated launchers)
            cn = int(rng.integers(1, 30))
            cpop = int(rng.integers(cn + 1, 10 * cn + 30))
            cme = int(rng.integers(cpop + 10, cpop + 2500))
            cxs, clo, chi = O.golden_x0(obj, cn)
            ckw = {}
            r = rng.random()
            if r < 0.25:
                ckw["ftol_rel"] = 10.0 ** -int(rng.integers(2, 8))
            elif r < 0.4:
                ckw["xtol_rel"] = 10.0 ** -int(rng.integers(2, 6))
            o = nlopt_amd.Opt(nlopt_amd.GN_CRS2_LM, cn)
            o.set_lower_bounds(clo); o.set_upper_bounds(chi); o.set_min_objective(nlopt_amd.objective(obj))
            o.set_population(cpop); o.set_maxeval(cme)
            if "ftol_rel" in ckw:
                o.set_ftol_rel(ckw["ftol_rel"])
            if "xtol_rel" in ckw:
                o.set_xtol_rel(ckw["xtol_rel"])
            wf = float(rng.choice([0.0, 0.5, 1.5, 3.0, 12.0]))
            ms = int(rng.choice([0, 0, 1, 2, 9, 200]))
            if wf:
                o.set_param("amd_window_factor", wf)
            if ms:
                o.set_param("amd_max_spec", ms)
            if rng.random() < 0.2:
                o.set_param("amd_host_eval", 1)
            o.set_comm(comm)
            o.enable_trace(cme + 64)
            nlopt_amd.srand(seed)
            x, minf, ret = o.optimize_raw(cxs)
            p = O.run_port_crs(obj, cn, cpop, seed, maxeval=cme, trace_cap=cme + 64, **ckw)
            t = o.trace()
            assert (ret, o.get_numevals(), minf) == (p["ret"], p["nevals"], p["minf"]), ("crs", draw, cn, cpop, wf, ms, ret, p["ret"], o.get_numevals(), p["nevals"])
            assert np.array_equal(x, p["x"]), ("crs x", draw)
            for key in ("f", "row", "kind", "accepted"):
                assert np.array_equal(t[key], p["trace"][key]), ("crs trace", key, draw, cn, cpop, wf, ms)
            assert L.nla_genrand_int32() == O.port().orc_genrand_int32(), ("crs stream position", draw)
            # ESCH
            en = int(rng.integers(1, 24))
            epop = int(rng.integers(0, 50))
            eme = int(rng.integers(50, 3000))
            exs, elo, ehi = O.golden_x0(obj, en)
            ekw = {}
            if rng.random() < 0.25:
                ekw["stopval"] = float(rng.uniform(0.1, 30.0))
            o = nlopt_amd.Opt(nlopt_amd.GN_ESCH, en)
            o.set_lower_bounds(elo); o.set_upper_bounds(ehi); o.set_min_objective(nlopt_amd.objective(obj))
            if epop:
                o.set_population(epop)
            o.set_maxeval(eme)
            if "stopval" in ekw:
                o.set_stopval(ekw["stopval"])
            if rng.random() < 0.2:
                o.set_param("amd_host_eval", 1)
            o.enable_trace(eme + 64)
            nlopt_amd.srand(seed)
            x, minf, ret = o.optimize_raw(exs)
            p = O.run_port_esch(obj, en, epop, seed, maxeval=eme, trace_cap=eme + 64, **ekw)
            t = o.trace()
            assert (ret, o.get_numevals(), minf) == (p["ret"], p["nevals"], p["minf"]), ("esch", draw, en, epop, ret, p["ret"], o.get_numevals(), p["nevals"])
            assert np.array_equal(x, p["x"]) and np.array_equal(t["f"], p["trace"]["f"]), ("esch trace", draw)
            assert L.nla_genrand_int32() == O.port().orc_genrand_int32(), ("esch stream position", draw)
            # ISRES
            pop = int(rng.integers(6, 70))
            ncon = int(rng.integers(0, 3)) if n >= 4 else 0
            neq = int(rng.integers(0, 2)) if n >= 4 else 0
            me = int(rng.integers(pop, 9 * pop))
            kw = {}
            if rng.random() < 0.3:
                kw["stopval"] = float(rng.uniform(0.5, 50.0))
            o = nlopt_amd.Opt(nlopt_amd.GN_ISRES, n)
            o.set_lower_bounds(lo); o.set_upper_bounds(hi); o.set_min_objective(nlopt_amd.objective(obj))
            o.set_population(pop); o.set_maxeval(me)
            if "stopval" in kw:
                o.set_stopval(kw["stopval"])
            if ncon:
                o.add_blocksum_constraints(ncon, 1e-8)
            if neq:
                o.add_blocksum_constraints(neq, 1e-8, True)
            o.set_comm(comm)
            o.enable_trace(me + 64)
            nlopt_amd.srand(seed)
            x, minf, ret = o.optimize_raw(xs)
            p = O.run_port_isres(obj, n, pop, seed, nineq=ncon, neq=neq, maxeval=me, **kw)
            t = o.trace()
            assert (ret, o.get_numevals(), minf) == (p["ret"], p["nevals"], p["minf"]), ("isres", draw, ret, p["ret"], o.get_numevals(), p["nevals"])
            assert np.array_equal(x, p["x"]) and np.array_equal(t["f"], p["ftrace"]), ("isres", draw)
            assert L.nla_genrand_int32() == O.port().orc_genrand_int32(), ("isres stream position", draw)
            # MLSL
            ns = int(rng.integers(0, 40))
            lds = bool(rng.random() < 0.5)
            local = ["lbfgs", "mma"][int(rng.integers(2))]
            me = int(rng.integers(200, 4000))
            tol = 10.0 ** -int(rng.integers(4, 10))
            lme = int(rng.integers(5, 60)) if rng.random() < 0.3 else 0
            o = nlopt_amd.Opt(nlopt_amd.G_MLSL_LDS if lds else nlopt_amd.G_MLSL, n)
            o.set_lower_bounds(lo); o.set_upper_bounds(hi); o.set_min_objective(nlopt_amd.objective(obj))
            loc = nlopt_amd.Opt(nlopt_amd.LD_MMA if local == "mma" else nlopt_amd.LD_LBFGS, n)
            loc.set_ftol_rel(tol)
            if lme:
                loc.set_maxeval(lme)
            L.nlopt_set_local_optimizer(o._h, loc._h)
            if ns:
                o.set_population(ns)
            o.set_maxeval(me)
            o.set_comm(comm)
            o.enable_trace(me + 4096)
            nlopt_amd.srand(seed)
            x, minf, ret = o.optimize_raw(xs)
            p = O.run_port_mlsl(obj, n, ns, seed, maxeval=me, local=local, lds=lds, local_ftol_rel=tol, local_maxeval=lme)
            t = o.trace()
            assert (ret, o.get_numevals(), minf) == (p["ret"], p["nevals"], p["minf"]), ("mlsl", draw, local, lds, ret, p["ret"], o.get_numevals(), p["nevals"])
            assert np.array_equal(x, p["x"]), ("mlsl", draw)
            locs = t[t["kind"] == 4]
            assert np.array_equal(locs["f"], p["floc"]) and np.array_equal(locs["accepted"], p["eloc"]), ("mlsl local searches", draw)
            assert L.nla_genrand_int32() == O.port().orc_genrand_int32(), ("mlsl stream position", draw)
            checked += 1
        res = dict(checked=np.array([checked]))
    else:
        raise SystemExit("unknown case " + case)
    if os.environ.get("NLA_TEST_MOCK_RCCL"):        # the same library instance comm.c dlopen()ed: how much of the run it carried
        import ctypes as C
        st = (C.c_long * 3)()
        C.CDLL(os.environ["NLA_RCCL_LIBRARY"]).mock_rccl_stats(st)
        res["rccl_calls"] = np.array(list(st))
    np.savez(out + ".rank%d.npz" % rank, **res)
    dist.barrier()
    if os.environ.get("NLA_TEST_SHM"):
        comm.destroy()                                # the last rank to leave removes the segment's name (comm.c shm_close)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
