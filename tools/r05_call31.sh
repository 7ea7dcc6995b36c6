#!/bin/bash
# Round 5, GPU call 31: host-side API calls over the last whole iteration of a config-4 run (10 ms before the end of the last searches' launch
# to 2 ms after): anything synchronous besides the waits?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c31; mkdir -p $O
timeout -k 5 100 rocprofv3 --hip-trace --kernel-trace -d $O/ka -o mlsl -- python bench.py --workload mlsl --steps 2 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/ka.err
f=$(find $O/ka -name '*.db' | head -1)
python - $f > $O/api_iter.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("select name, start, end from regions order by start"))
k = list(cur.execute("select start, end, name from kernels order by start"))
lb = [r for r in k if "lbfgs_resident" in r[2]]
t_end = lb[-1][1]
print("# t = 0: end of the last lbfgs_resident_kernel; API calls except the polling ones, and kernels on the main stream's critical path; us")
skip = ("hipStreamQuery", "hipGetLastError", "__hipPushCallConfiguration", "__hipPopCallConfiguration", "hipEventQuery")
ev = [((st - t_end) / 1e3, (en - st) / 1e3, "API  " + nm) for nm, st, en in rows if t_end - 10.5e6 <= st <= t_end + 2.0e6 and nm not in skip]
ev += [((st - t_end) / 1e3, (en - st) / 1e3, "KERN " + nm.split("(")[0]) for st, en, nm in k if t_end - 10.5e6 <= st <= t_end + 2.0e6 and "copyBuffer" not in nm]
for t, d, nm in sorted(ev):
    print("%10.1f %9.1f  %s" % (t, d, nm))
PY
rm -rf $O/ka; wc -l $O/api_iter.txt
