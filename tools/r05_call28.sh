#!/bin/bash
# Round 5, GPU call 28: host-side API timeline of config 4 (what the host does between the pairs' arrival and the column-minimum launch)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c28; mkdir -p $O
timeout -k 5 300 rocprofv3 --hip-trace --kernel-trace -d $O/ka -o mlsl -- python bench.py --workload mlsl --steps 2 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/ka.err
f=$(find $O/ka -name '*.db' | head -1)
python - $f > $O/api_tail.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("select name, start, end from regions order by start"))
k = list(cur.execute("select start, end, name from kernels order by start"))
# the last lbfgs_resident launch: print every API call from 1 ms before its END to 4 ms after, and the kernels of that span
lb = [r for r in k if "lbfgs_resident" in r[2]]
t_end = lb[-1][1]
print("# t = 0: end of the last lbfgs_resident_kernel")
ev = [((st - t_end) / 1e3, (en - st) / 1e3, "API  " + nm) for nm, st, en in rows if t_end - 1.0e6 <= st <= t_end + 4.0e6]
ev += [((st - t_end) / 1e3, (en - st) / 1e3, "KERN " + nm.split("(")[0]) for st, en, nm in k if t_end - 1.0e6 <= st <= t_end + 4.0e6]
for t, d, nm in sorted(ev):
    print("%10.1f %9.1f  %s" % (t, d, nm))
PY
rm -rf $O/ka; wc -l $O/api_tail.txt; tail -1 $O/bench.json | cut -c1-200
