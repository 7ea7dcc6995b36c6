"""Turn rocprofv3's rocpd sqlite output (ROCm 7.2 default) into the text summaries committed here.
usage: python profiles/summarize_rocpd.py <results.db> [--pmc]"""
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
    rows = list(cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                            "from kernels group by name order by sum(end-start) desc"))
    tot = sum(r[2] for r in rows)
    print("kernel,calls,total_ns,avg_ns,min_ns,max_ns,percent")
    for r in rows:
        print("%s,%d,%d,%.1f,%d,%d,%.2f" % (r[0].split("(")[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot))


if __name__ == "__main__":
    main()
