#!/bin/bash
# Round 5, GPU call 29: host-side API calls of the steady state of config 3 (ISRES) and config 2 (CRS2_LM n = 512): anything that allocates,
# frees or copies synchronously inside a generation / a window?  (config 4's trace found a 0.57 ms hipHostFree per iteration, call 28)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c29; mkdir -p $O
probe() {   # probe <name> <bench args...>
  local name=$1; shift
  timeout -k 5 200 rocprofv3 --hip-trace --kernel-trace -d $O/k_$name -o $name -- python bench.py "$@" > $O/bench_$name.json 2> $O/k_$name.err
  f=$(find $O/k_$name -name '*.db' | head -1)
  python - $f $name > $O/api_$name.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("select name, start, end from regions order by start"))
k = list(cur.execute("select start, end, name from kernels order by start"))
# the last third of the kernels' time span = the timed steps
t0, t1 = k[0][0], k[-1][1]
lo = t0 + (t1 - t0) * 0.55
hi = t0 + (t1 - t0) * 0.95
agg = {}
for nm, st, en in rows:
    if lo <= st <= hi:
        a = agg.setdefault(nm, [0, 0, 0]); a[0] += 1; a[1] += en - st; a[2] = max(a[2], en - st)
print("# %s: host-side API calls between 55 %% and 95 %% of the run's kernel span (%.1f ms)" % (sys.argv[2], (hi - lo) / 1e6))
print("function,calls,total_us,avg_us,max_us")
for nm, (c, tot, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print("%s,%d,%.1f,%.2f,%.1f" % (nm, c, tot / 1e3, tot / 1e3 / c, mx / 1e3))
print("# calls longer than 100 us in that span (start offset ms, duration us, name)")
for nm, st, en in rows:
    if lo <= st <= hi and en - st > 100e3 and "Synchronize" not in nm:
        print("%.2f,%.1f,%s" % ((st - lo) / 1e6, (en - st) / 1e3, nm))
PY
  rm -rf $O/k_$name; head -30 $O/api_$name.txt
}
probe isres --workload isres --steps 3 --warmup 1 --no-cpu-baseline
probe n512 --headline-only --no-cpu-baseline --obj rastrigin --n 512 --steps 6 --warmup 1 --evals-per-step 20000
