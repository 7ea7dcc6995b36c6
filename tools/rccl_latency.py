"""What a call of the library's RCCL all-gather costs on this box (development aid, MI355X): a 1-rank communicator (one GPU per box: the
data movement is a local copy, what is measured is RCCL's launch path — the per-call floor a pass of the column-sharded CRS2_LM pays),
device buffers of several sizes, (a) call + stream synchronisation each time, (b) back to back with one synchronisation; and the
host-data variant (H2D + all-gather + D2H + synchronisation) of 8 bytes, which the stop agreement used per pass before it rode on the
candidates' all-gather.  profiles/r03_rccl_latency.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nlopt_amd  # noqa: E402

L = nlopt_amd.lib()
c = nlopt_amd.Comm.rccl(0, 1, nlopt_amd.rccl_unique_id())
st = L.nla_stream_create()
print("RCCL all-gather through nla_comm_allgather_dev, 1-rank communicator, one MI355X")
for nbytes in (16, 4096, 65536, 524288, 4194304):
    src, dst = nlopt_amd.DevBuf(nbytes), nlopt_amd.DevBuf(nbytes)
    for _ in range(20):
        L.nla_comm_allgather_dev(c._h, src.ptr, dst.ptr, nbytes, st)
    L.nla_stream_sync(st)
    reps = 300
    t0 = time.perf_counter()
    for _ in range(reps):
        L.nla_comm_allgather_dev(c._h, src.ptr, dst.ptr, nbytes, st)
        L.nla_stream_sync(st)
    t_sync = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        L.nla_comm_allgather_dev(c._h, src.ptr, dst.ptr, nbytes, st)
    L.nla_stream_sync(st)
    t_b2b = (time.perf_counter() - t0) / reps
    print("  %8d bytes: call + sync %.1f us   back to back %.1f us per call" % (nbytes, 1e6 * t_sync, 1e6 * t_b2b))
a = np.arange(1, dtype=np.float64)
for _ in range(20):
    c.allgather_host(a)
t0 = time.perf_counter()
for _ in range(300):
    c.allgather_host(a)
print("  host data, 8 bytes (H2D + all-gather + D2H + sync, through ctypes): %.1f us per call" % (1e6 * (time.perf_counter() - t0) / 300))
c.destroy()
