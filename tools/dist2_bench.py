"""time of mlsl_dist2_kernel alone (events on the launch stream) at the shapes of config 4's iterations:
    python tools/dist2_bench.py            (NLOPT_AMD_LIB=<variant> for an A/B)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nlopt_amd  # noqa: E402

L = nlopt_amd.lib()
L.nla_k_mlsl_dist2.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
for f in ("nla_event_create", "nla_stream_create"):
    getattr(L, f).restype = C.c_void_p
L.nla_event_record.argtypes = [C.c_void_p, C.c_void_p]
L.nla_event_sync.argtypes = [C.c_void_p]
L.nla_event_elapsed_ms.argtypes = [C.c_void_p, C.c_void_p]
L.nla_event_elapsed_ms.restype = C.c_float
L.nla_stream_sync.argtypes = [C.c_void_p]
st = L.nla_stream_create()
e0, e1 = L.nla_event_create(), L.nla_event_create()
rng = np.random.default_rng(1)
for n, na, nb in [(4096, 1000, 300), (4096, 305, 300), (4096, 305, 4000), (4096, 1000, 1800), (4096, 305, 8000), (4096, 1000, 4000), (4096, 1000, 8000), (512, 1000, 4000), (64, 1000, 16000)]:
    ld = (n + 1) & ~1
    dA = nlopt_amd.DevBuf.from_array(rng.uniform(-30, 30, (na, ld)))
    dB = nlopt_amd.DevBuf.from_array(rng.uniform(-30, 30, (nb, ld)))
    dD = nlopt_amd.DevBuf(8 * na * nb)
    for _ in range(2):
        L.nla_k_mlsl_dist2(n, ld, dA.ptr, na, dB.ptr, nb, dD.ptr, st)
    L.nla_event_record(e0, st)
    reps = 10
    for _ in range(reps):
        L.nla_k_mlsl_dist2(n, ld, dA.ptr, na, dB.ptr, nb, dD.ptr, st)
    L.nla_event_record(e1, st)
    L.nla_event_sync(e1)
    ms = L.nla_event_elapsed_ms(e0, e1) / reps
    ops = 3.0 * na * nb * n
    print("dist2 n=%d %d x %d: %.3f ms  %.1f T fp64 op/s (%.2f of the 39.3 T/s of unfused fp64 add/mul)" % (n, na, nb, ms, ops / ms / 1e9, ops / ms / 1e9 / 39.3))
