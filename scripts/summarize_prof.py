#!/usr/bin/env python3
"""Turn rocprofv3 rocpd (.db) outputs under gpurun_out/ into the text summaries committed under profiles/.

    python scripts/summarize_prof.py <title> <trace.db or ''> [pmc.db ...]
"""
import sqlite3
import sys


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def main():
    title, trace, pmcs = sys.argv[1], sys.argv[2], sys.argv[3:]
    print("# " + title)
    if trace:
        cur = sqlite3.connect(trace).cursor()
        print("# rocprofv3 --kernel-trace --stats (durations in microseconds)")
        print("%-48s %6s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        for name, calls, tot, avg, pct in cur.execute("select * from top_kernels"):
            print("%-48s %6d %12.1f %12.1f %6.2f%%" % (short(name)[:48], calls, tot, avg, pct))
    for db in pmcs:
        cur = sqlite3.connect(db).cursor()
        print("\n# rocprofv3 --pmc pass %s (separate run; mean per dispatch, summed over the chip)" % db.split("/")[-2])
        q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
             "where kernel_name like '%k_scan%' or kernel_name like '%k_tail%' or kernel_name like '%k_cluster%' or kernel_name like '%k_rgb_to_gray%' or kernel_name like '%k_puploc%' group by kernel_name, counter_name")
        for k, c, n, avg in cur.execute(q):
            print("%-40s %-30s n=%3d mean=%.6g" % (short(k)[:40], c, n, avg))


if __name__ == "__main__":
    main()
