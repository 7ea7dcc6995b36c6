"""development aid: summarise a NLA_CRS_PASS_LOG file (n,K,nW,done_in,fresh,stopped,rows,ms per pass of crs_advance_kernel)"""
import sys
import numpy as np

a = np.loadtxt(sys.argv[1], delimiter=",", ndmin=2)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
a = a[a[:, 0] == n]
K, nW, done_in, fresh, stopped, rows, ms = a[:, 1], a[:, 2], a[:, 3], a[:, 4], a[:, 5], a[:, 6], a[:, 7]
eq = rows / (n + 1.0)                       # work of the pass in fresh-slot equivalents
gb = rows * 8.0 * n / 1e9
print("passes %d   K mean %.1f   already complete %.1f   fresh %.2f   stopped short %.2f   work %.2f slot-equivalents   %.1f us   %.0f GB/s overall"
      % (len(a), K.mean(), done_in.mean(), fresh.mean(), stopped.mean(), eq.mean(), 1e3 * ms.mean(), gb.sum() / ms.sum() * 1e3))
print("work (slot-eq)  passes   mean us   GB/s    share of time")
edges = [0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 100]
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (eq >= lo) & (eq < hi)
    if m.any():
        print("%5.0f-%-5.0f   %7d   %7.1f   %6.0f   %5.1f %%" % (lo, hi, m.sum(), 1e3 * ms[m].mean(), gb[m].sum() / ms[m].sum() * 1e3, 100 * ms[m].sum() / ms.sum()))
