"""What does a pass of the column-sharded CRS2_LM cost?  (development aid, run on the MI355X box)

The GPU box has ONE GPU, so the ranks of a sharded job share it (round 4: over the library's own shared-memory transport, comm.c —
round 3's figure of 226 ms per pass was the Python / gloo hop of the test transport, not the library): the gather of a pass is not faster
than on one rank — what this measures is everything else a sharded pass adds (the candidates' pack / all-gather / evaluation, the
stop agreement, the pointer-based list upload) next to the single-process conservative passes and the chain kernel, at the metric
configuration and at BASELINE config 5's population.  profiles/r03_shard_probe.txt; DESIGN.md section 6 builds its 8-GPU
estimate on these numbers."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _mp_launch import run_world  # noqa: E402

n = 4096
for pop, evals in ((100000, 6000),):
    base = dict(obj="griewank", n=n, pop=pop, seed=42, evals=evals)
    rows = []
    SHM = dict(NLA_TEST_SHM="1")
    # round 6: windows resolved on the device on the column-sharded population ("amd_cu_share" = k: each process on every k-th compute
    # unit, so that k ranks can share the one GPU of this box) next to ONE rank confined to the same share — the difference is what the
    # in-launch exchange costs, the gather per rank is 1 / world of the rows on 1 / world of the chip
    for label, world, params, env in (("1 rank, chain kernel (default)", 1, {}, {}),
                                      ("1 rank, chain kernel on 1/2 of the CUs", 1, {"amd_cu_share": 2}, {}),
                                      ("2 ranks on 1/2 of the CUs each, column-sharded WINDOWS (round 6)", 2, {"amd_cu_share": 2}, SHM),
                                      ("1 rank, chain kernel on 1/4 of the CUs", 1, {"amd_cu_share": 4}, {}),
                                      ("4 ranks on 1/4 of the CUs each, column-sharded WINDOWS (round 6)", 4, {"amd_cu_share": 4}, SHM),
                                      ("8 ranks on 1/8 of the CUs each, column-sharded WINDOWS (round 6)", 8, {"amd_cu_share": 8}, SHM),
                                      ("1 rank, chain kernel on 1/8 of the CUs", 1, {"amd_cu_share": 8}, {}),
                                      ("1 rank, conservative passes", 1, {"amd_forward": 0}, {}),
                                      ("2 ranks sharing the GPU, column-sharded conservative passes, shm transport (comm.c)", 2, {"amd_shard_windows": 0}, SHM)):
        try:
            res = run_world("gpu_crs_rate", dict(base, params=params), world=world, timeout=600, extra_env=env)
            d = res[0]
            rows.append(dict(case=label, pop=pop, evals_per_s=float(d["evals_per_s"][0]), passes=int(d["passes"][0]), us_per_pass=1e6 * float(d["dt"][0]) / max(int(d["passes"][0]), 1),
                             evals_per_pass=float(d["evals"][0]) / max(int(d["passes"][0]), 1), t_init_s=float(d["t_init"][0]),
                             gather_ms_per_timed_pass=float(d["gather_ms"][0]) / max(int(d["gather_launches"][0]), 1),
                             allgather_MB=float(d["allgather_bytes"][0]) / 1e6, minf=float(d["minf"][0]),
                             step_ret=[int(v) for v in d["step_ret"]], errmsg=str(d["errmsg"])))
        except Exception as e:
            rows.append(dict(case=label, pop=pop, error=repr(e)[:300]))
        print(json.dumps(rows[-1]), flush=True)
