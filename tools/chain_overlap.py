"""chain_overlap.py <results.db> — per-launch durations of crs_chain_kernel from a rocprofv3 kernel trace (rocpd sqlite), split by
which other kernels were running on the device at the same time (development / evidence tool)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
chain = [(s, e) for nm, s, e in rows if "crs_chain" in nm]
others = [(nm.split("(")[0].split("<")[0].replace("void ", ""), s, e) for nm, s, e in rows if "crs_chain" not in nm and e - s > 20000]
buckets = {}
for s, e in chain:
    tags = sorted({nm for nm, os_, oe in others if os_ < e and oe > s and min(e, oe) - max(s, os_) > 0.3 * (e - s)})
    buckets.setdefault("+".join(tags) or "alone", []).append(e - s)
print("crs_chain_kernel launches: %d" % len(chain))
for k, v in sorted(buckets.items(), key=lambda kv: -len(kv[1])):
    v.sort()
    print("%-50s n=%4d  mean %8.1f us  median %8.1f  min %8.1f  max %8.1f" % (k, len(v), sum(v) / len(v) / 1e3, v[len(v) // 2] / 1e3, v[0] / 1e3, v[-1] / 1e3))
gaps = [chain[i + 1][0] - chain[i][1] for i in range(len(chain) - 1)]
gaps.sort()
print("gap between consecutive chain launches: median %.1f us, mean %.1f us, p90 %.1f us" % (gaps[len(gaps) // 2] / 1e3, sum(gaps) / len(gaps) / 1e3, gaps[int(0.9 * len(gaps))] / 1e3))
