"""the ISRES ranking pipeline's unit costs on the device (development aid): pop elements through U chained units of 64 sweeps each
   time(1 unit) / (pop + 126)            = ns per tick of a unit that never waits for its upstream (the input stream is complete)
   (time(U) - time(1)) / (U - 1)         = what every further unit adds to the end-to-end time (its start lag behind its predecessor)
   python tools/sr_unit_bench.py [pop]"""
import sys, ctypes as C, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import nlopt_amd
from nlopt_amd import DevBuf
L = nlopt_amd.lib()
pop = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
rng = np.random.default_rng(3)
f = rng.random(pop); pen = np.where(rng.random(pop) < 0.3, 0.0, rng.random(pop))
roww = (pop - 1 + 63) // 64
L.nla_event_create.restype = C.c_void_p
L.nla_event_record.argtypes = [C.c_void_p, C.c_void_p]; L.nla_event_elapsed_ms.argtypes = [C.c_void_p, C.c_void_p]; L.nla_event_elapsed_ms.restype = C.c_float
L.nla_event_sync.argtypes = [C.c_void_p]
e0, e1 = L.nla_event_create(), L.nla_event_create()
res = {}
for U in (1, 2, 8, 32, 128):
    ns = 64 * U
    dF, dP = DevBuf.from_array(f), DevBuf.from_array(pen)
    dstreams, dsorted = DevBuf(8 * (U + 1) * pop), DevBuf(4 * pop)
    dbits = DevBuf.from_array(rng.integers(0, 2**63, size=ns * roww, dtype=np.int64).astype(np.uint64))
    dsw, dirank = DevBuf(pop), DevBuf(4 * pop)
    best = 1e9
    for rep in range(3):
        prog = np.zeros(U + 1, np.int32); prog[0] = pop
        dprog, dticket = DevBuf.from_array(prog), DevBuf.from_array(np.zeros(1, np.int32))
        assert L.nla_k_isres_rank_count(pop, dF.ptr, dP.ptr, dstreams.ptr, dsorted.ptr, None) == 0
        assert L.nla_stream_sync(None) == 0
        L.nla_event_record(e0, None)
        assert L.nla_k_isres_stochrank(pop, ns, dstreams.ptr, dprog.ptr, dbits.ptr, dticket.ptr, dsw.ptr, dirank.ptr, None) == 0
        L.nla_event_record(e1, None)
        assert L.nla_stream_sync(None) == 0
        best = min(best, L.nla_event_elapsed_ms(e0, e1))
    res[U] = best
    print("units %4d: %8.3f ms" % (U, best), flush=True)
t1 = res[1]
print("pop %d: one unit %.3f ms = %.1f ns per tick (pop + 126 ticks); each further unit adds %.2f us (U=8) %.2f us (U=32) %.2f us (U=128) = %.0f ticks of that length"
      % (pop, t1, 1e6 * t1 / (pop + 126), 1e3 * (res[8] - t1) / 7, 1e3 * (res[32] - t1) / 31, 1e3 * (res[128] - t1) / 127, (res[128] - t1) / 127 / (t1 / (pop + 126))))
