"""chain_latency.py — what the pieces of crs_chain_kernel cost on this box (development / evidence tool).

One window of K blocks at the metric shape (Griewank n = 4096, 100 000 rows) is launched with the worst-row list arranged
so that the kernel's three regimes are timed apart:
  free      nW = 0: no pick is a hazard, nothing waits — gather + evaluation only
  accept    every trial is accepted (Wf = +huge): a pick of W[j] waits for producer j and reads its TX
  reject    nothing is accepted (Wf = -huge): a pick of W[j] waits until every block before the slot is resolved
run on the GPU box:  python tools/chain_latency.py [n] [N]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np

import nlopt_amd
from nlopt_amd import DevBuf

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
L = nlopt_amd.lib()
ld = (n + 15) & ~15
rng = np.random.default_rng(5)
oid = nlopt_amd.OBJECTIVES["griewank"]
lo, hi = nlopt_amd.objective_box("griewank")
lb, ub = np.full(ld, lo), np.full(ld, hi)
blk_rows = 1000
blk = np.zeros((blk_rows, ld))
blk[:, :n] = rng.uniform(lo, hi, size=(blk_rows, n))
dX = DevBuf(8 * ld * (N + 8))
for r0 in range(0, N, blk_rows):
    cnt = min(blk_rows, N - r0)
    assert L.nla_memcpy_h2d(dX.ptr + 8 * ld * r0, blk.ctypes.data, 8 * ld * cnt, None) == 0
assert L.nla_stream_sync(None) == 0
KMAX = 256
ring = 2 * KMAX + 3
i0 = 7
pos = np.zeros((ring, n), np.int32)
for b in range(ring):
    pos[b] = np.sort(rng.choice(N - 2, n, replace=False))
jn = rng.integers(0, n, ring).astype(np.int32)
last = np.zeros(ring, np.int32)
w = rng.integers(0, 2 ** 32, 2 * n * ring, dtype=np.uint64).astype(np.uint32)
mask = 511


class St(C.Structure):
    _fields_ = [("fT", C.c_double), ("fM", C.c_double), ("t", C.c_int32), ("pad", C.c_int32)]


dlb, dub, dw = DevBuf.from_array(lb), DevBuf.from_array(ub), DevBuf.from_array(w)
dj, dp, dl = DevBuf.from_array(jn), DevBuf.from_array(pos), DevBuf.from_array(last)
dTX, dTM = DevBuf(8 * ld * (mask + 1), uncached=True), DevBuf(8 * ld * (mask + 1), uncached=True)
cb = L.nla_crs_chain_ctrl_bytes(256, 256)
dctrl = DevBuf.from_array(np.zeros(cb, np.uint8), uncached=True)
dst = DevBuf(C.sizeof(St) * KMAX)
fwcap = 48
dcnt, drec = DevBuf(4 * KMAX), DevBuf(4 * KMAX * fwcap)
chunks = L.nla_crs_chain_chunks(n, ld)
e0, e1 = L.nla_event_create(), L.nla_event_create()
L.nla_event_elapsed_ms.restype = C.c_float
L.nla_event_elapsed_ms.argtypes = [C.c_void_p, C.c_void_p]
L.nla_event_record.argtypes = [C.c_void_p, C.c_void_p]
L.nla_event_sync.argtypes = [C.c_void_p]
ticket = 0
bytes_per_trial = 8.0 * n * (n + 1) + 8.0 * n * 4          # the gather + the evaluation's passes over the point
print("n = %d, N = %d, %d chunks per slot, %.1f MB per trial" % (n, N, chunks, bytes_per_trial / 1e6))
for K in ((8, 24, 48, 96, 192) if n >= 2048 else (32, 64, 128, 256)):
    for mode in ("free", "accept", "reject"):
        nW = 0 if mode == "free" else min(K, 256)
        W = rng.choice(N - 2, max(nW, 1), replace=False).astype(np.int64) + 8
        Wf = np.full(max(nW, 1), 1e300 if mode == "accept" else -1e300)
        dW, dWf = DevBuf.from_array(W), DevBuf.from_array(Wf)
        best = 1e30
        for rep in range(4):
            first = 3 * ring + 2 + rep * K
            L.nla_event_record(e0, None)
            rc = L.nla_k_crs_chain(oid, n, ld, dX.ptr, i0, -1e300, dj.ptr, dp.ptr, dl.ptr, dw.ptr, ring, first, K, dW.ptr, dWf.ptr, nW, 0, mask,
                                   dlb.ptr, dub.ptr, dTX.ptr, dTM.ptr, dctrl.ptr, ticket, dst.ptr, dcnt.ptr, drec.ptr, fwcap, None)
            assert rc == 0, rc
            L.nla_event_record(e1, None)
            L.nla_event_sync(e1)
            ticket = (ticket + L.nla_crs_chain_tickets(n, ld, K)) & 0xffffffff
            ms = L.nla_event_elapsed_ms(e0, e1)
            best = min(best, ms)
        cnt = dcnt.to_array(np.uint32, K)
        print("K = %3d  %-6s  %8.3f ms  %6.1f us/slot  %7.1f GB/s   hazard picks per slot: mean %.1f" %
              (K, mode, best, 1e3 * best / K, K * bytes_per_trial / 1e6 / best, cnt.mean()))
