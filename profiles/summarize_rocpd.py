"""Turn rocprofv3's rocpd sqlite output (ROCm 7.2 default) into the text summaries committed here.
usage: python profiles/summarize_rocpd.py <results.db> [--pmc | --timeline [first_row [rows]]]
--timeline: every kernel dispatch in start order — start offset (us), duration (us), gap to the previous END on any queue (us),
queue / stream ids where the view has them, name — what the per-kernel sums cannot show: idle gaps and overlap."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    if "--pmc" in sys.argv:
        cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        q = ("select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection "
             "group by kernel_name, counter_name order by sum(value) desc")
        print("# columns of counters_collection:", cols)
        print("kernel,counter,dispatches,sum,avg_per_dispatch")
        for r in cur.execute(q):
            print("%s,%s,%d,%.6g,%.6g" % (r[0].split("(")[0], r[1], r[2], r[3], r[4]))
        return
    if "--api" in sys.argv:
        # host-side API calls (rocprofv3 --hip-trace): per function count / total / average, then the calls in start order with the host
        # time between the END of one call and the START of the next (what the application itself spent)
        names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
        cand = [t for t in names if t.lower() in ("regions", "hip_api", "api", "regions_and_samples")] or [t for t in names if "region" in t.lower()]
        print("# tables/views:", names)
        for t in cand[:1]:
            cols = [r[1] for r in cur.execute("pragma table_info(%s)" % t)]
            print("# %s columns:" % t, cols)
            if not {"name", "start", "end"} <= set(cols):
                continue
            rows = list(cur.execute("select name, start, end from %s order by start" % t))
            agg = {}
            for nm, st, en in rows:
                a = agg.setdefault(nm, [0, 0])
                a[0] += 1; a[1] += en - st
            print("function,calls,total_us,avg_us")
            for nm, (c, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
                print("%s,%d,%.1f,%.2f" % (nm, c, tot / 1e3, tot / 1e3 / c))
            lim = int(sys.argv[sys.argv.index("--api") + 1]) if len(sys.argv) > sys.argv.index("--api") + 1 else 0
            if lim:
                mid = len(rows) // 2
                print("idx,start_us,dur_us,host_gap_before_us,name")
                last = rows[mid][1]
                for k, (nm, st, en) in enumerate(rows[mid:mid + lim]):
                    print("%d,%.1f,%.1f,%.1f,%s" % (k, (st - rows[mid][1]) / 1e3, (en - st) / 1e3, (st - last) / 1e3, nm))
                    last = en
        return
    if "--timeline" in sys.argv:
        i = sys.argv.index("--timeline")
        first = int(sys.argv[i + 1]) if len(sys.argv) > i + 1 else 0
        count = int(sys.argv[i + 2]) if len(sys.argv) > i + 2 else 100000
        cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
        extra = [c for c in ("queue_id", "stream_id", "grid_x", "workgroup_x", "grid_size", "workgroup_size") if c in cols]
        rows = list(cur.execute("select start, end, name%s from kernels order by start" % "".join(", " + c for c in extra)))
        print("# columns of kernels:", cols)
        print("idx,start_us,dur_us,gap_us,%sname" % "".join(c + "," for c in extra))
        t0, last_end = (rows[0][0], rows[0][0]) if rows else (0, 0)
        for k, r in enumerate(rows):
            if first <= k < first + count:
                print("%d,%.1f,%.1f,%.1f,%s%s" % (k, (r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, (r[0] - last_end) / 1e3,
                                                  "".join(str(v) + "," for v in r[3:]), r[2].split("(")[0]))
            last_end = max(last_end, r[1])
        return
    rows = list(cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                            "from kernels group by name order by sum(end-start) desc"))
    tot = sum(r[2] for r in rows)
    print("kernel,calls,total_ns,avg_ns,min_ns,max_ns,percent")
    for r in rows:
        print("%s,%d,%d,%.1f,%d,%d,%.2f" % (r[0].split("(")[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot))


if __name__ == "__main__":
    main()
