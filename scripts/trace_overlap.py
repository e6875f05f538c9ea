#!/usr/bin/env python3
"""Kernel begin/end timestamps of one rocprofv3 --kernel-trace run (rocpd .db): how long the big scales' launches (side stream)
really run next to the region launches, per step.

    python scripts/trace_overlap.py <trace.db> [steps_to_print]
"""
import sqlite3
import sys


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]


def main():
    db = sys.argv[1]
    nprint = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    con = sqlite3.connect(db)
    cur = con.cursor()
    rows = None
    for q in ("select name, start, end from kernels order by start",
              "select kernel_name, start, end from kernels order by start",
              "select name, start_timestamp, end_timestamp from kernels order by start_timestamp"):
        try:
            rows = cur.execute(q).fetchall()
            break
        except sqlite3.Error:
            continue
    if rows is None:
        names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
        print("no usable kernel table; objects:", names)
        for n in names:
            if "kernel" in n.lower():
                print(n, [c[1] for c in cur.execute(f"pragma table_info('{n}')")])
        return 1
    rows = [(short(n), s, e) for n, s, e in rows]
    # a step = from one k_scan_region launch pair to k_restore_order
    steps, curstep = [], []
    for n, s, e in rows:
        if not n.startswith(("k_scan", "k_tail", "k_restore", "k_big")):
            continue
        curstep.append((n, s, e))
        if n == "k_restore_order":
            steps.append(curstep)
            curstep = []
    print(f"# {db}: {len(rows)} dispatches, {len(steps)} scan steps")
    side = ("k_scan_big", "k_big_pool", "k_tail_deep", "k_scan_tile")
    tot = []
    for st in steps:
        t0 = min(s for _, s, _ in st)
        reg = [(s, e) for n, s, e in st if n == "k_scan_region"]
        sd = [(s, e) for n, s, e in st if n in side]
        reg_span = (min(s for s, _ in reg), max(e for _, e in reg)) if reg else (t0, t0)
        sd_span = (min(s for s, _ in sd), max(e for _, e in sd)) if sd else (t0, t0)
        ov = max(0, min(reg_span[1], sd_span[1]) - max(reg_span[0], sd_span[0]))
        end = max(e for _, _, e in st)
        tot.append(((end - t0) / 1e3, (reg_span[1] - reg_span[0]) / 1e3, (sd_span[1] - sd_span[0]) / 1e3, ov / 1e3,
                    sum(e - s for n, s, e in st) / 1e3))
    for i, st in enumerate(steps[-nprint:]):
        t0 = min(s for _, s, _ in st)
        print(f"## step {len(steps) - nprint + i}: begin / end (us after the step's first kernel start), duration")
        for n, s, e in st:
            print(f"  {n:<18s} {(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:9.1f}")
    if tot:
        k = tot[len(tot) // 2:]
        m = [sum(x[j] for x in k) / len(k) for j in range(5)]
        print(f"# mean over the last {len(k)} steps (us): step span {m[0]:.1f}, region span {m[1]:.1f}, side span {m[2]:.1f}, "
              f"overlap interval {m[3]:.1f}, sum of kernel durations {m[4]:.1f}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
