"""development / evidence tool: how long does crs_advance_kernel take as a function of the number of FRESH slots in the
launch (each a full n-row gather-sum, no hazards)?  K = 1 is the serial floor of one long item (32 workgroups, the memory
system idle); the growth with K shows where bandwidth takes over.  Usage: python tools/advance_latency.py [n] [N] [variant]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import nlopt_amd  # noqa: E402
import _oracle as O  # noqa: E402
from nlopt_amd import DevBuf  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
variant = int(sys.argv[3]) if len(sys.argv) > 3 else 0
L, P = nlopt_amd.lib(), O.port()
KMAX = 32
ld = (n + 1) & ~1
ring = 64
rng = np.random.default_rng(1)
w = rng.integers(0, 2 ** 32, size=2 * n * ring, dtype=np.uint32)
jn, pos, last = np.zeros(ring, np.int32), np.zeros(ring * n, np.int32), np.zeros(ring, np.int32)
P.orc_k_vitter.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
P.orc_k_vitter(n, N, w.ctypes.data, ring, jn.ctypes.data, pos.ctypes.data, last.ctypes.data)
dX = DevBuf(8 * ld * N)
L.nla_memset.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
L.nla_memset(dX.ptr, 0, 8 * ld * N, None)
lb, ub = np.full(n, -600.0), np.full(n, 600.0)
dlb, dub = DevBuf.from_array(lb), DevBuf.from_array(ub)
dj, dp, dl = DevBuf.from_array(jn), DevBuf.from_array(pos), DevBuf.from_array(last)
dW = DevBuf.from_array(np.zeros(KMAX, np.int64))
dTX = DevBuf(8 * ld * 64)
dt0, dt1 = DevBuf.from_array(np.zeros(KMAX, np.int32)), DevBuf(4 * KMAX)
L.nla_event_create.restype = C.c_void_p
L.nla_event_record.argtypes = [C.c_void_p, C.c_void_p]
L.nla_event_sync.argtypes = [C.c_void_p]
L.nla_event_elapsed_ms.argtypes = [C.c_void_p, C.c_void_p]
L.nla_event_elapsed_ms.restype = C.c_float
L.nla_k_crs_advance.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_int,
                                C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
e0, e1 = L.nla_event_create(), L.nla_event_create()
print("crs_advance_kernel, n=%d N=%d variant=%d: K fresh slots, no hazards; bytes = K * n * (n+1) * 8" % (n, N, variant))
for K in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
    best, tot, reps = 1e9, 0.0, 12
    for r in range(reps):
        first = (r * K) % (ring - K + 1)                 # different pick lists every repetition
        L.nla_event_record(e0, None)
        rc = L.nla_k_crs_advance(n, ld, dX.ptr, 0, dj.ptr, dp.ptr, dl.ptr, ring, first, K, dW.ptr, 0, dt0.ptr, dt1.ptr, 63,
                                 dlb.ptr, dub.ptr, dTX.ptr, variant, None)
        assert rc == 0
        L.nla_event_record(e1, None)
        L.nla_event_sync(e1)
        ms = L.nla_event_elapsed_ms(e0, e1)
        if r:
            best = min(best, ms)
            tot += ms
    avg = tot / (reps - 1)
    gb = K * n * (n + 1) * 8 / 1e9
    print("K=%2d  workgroups=%4d  avg %.3f ms  best %.3f ms  %.0f GB/s (avg)  %.0f GB/s (best)" % (K, K * ((n + 127) // 128), avg, best, gb / avg * 1e3, gb / best * 1e3), flush=True)
