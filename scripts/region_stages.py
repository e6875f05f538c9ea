#!/usr/bin/env python3
"""Per-stage cost of k_scan_region by ELIMINATION (debug library, PIGO_REG_CUT): reads the rocprofv3 outputs that
scripts/gpu_r5_region_model.sh leaves under <dir>/g<group>_cut<c>_{trace,pmc}/ and prints, per scale group, the kernel's duration and
SQ counters for every cut and the differences between consecutive cuts = what each stage adds.

    python scripts/region_stages.py <dir> <frames per launch>

cut 1 = copy + tables + stage 0 (tree 0, every window); 2 = + the second chunk stage; 3 = + the remaining chunk stages; 4 = + the
pool (trees up to the hand-over); 0 = the whole kernel (+ deep list)."""
import glob
import sqlite3
import sys

CUTS = [1, 2, 3, 4, 0]
NAMES = {1: "copy + stage 0", 2: "+ chunk stage 1", 3: "+ chunk stages 2..", 4: "+ pool", 0: "+ deep list (= whole kernel)"}
CTRS = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "GRBM_GUI_ACTIVE"]


def one(db, q):
    cur = sqlite3.connect(db).cursor()
    return list(cur.execute(q))


def main():
    root, frames = sys.argv[1], int(sys.argv[2])
    for g in (0, 1):
        rows = {}
        for c in CUTS:
            t = glob.glob(f"{root}/g{g}_cut{c}_trace/**/*.db", recursive=True)
            p = glob.glob(f"{root}/g{g}_cut{c}_pmc/**/*.db", recursive=True)
            if not t or not p:
                continue
            r = one(t[0], "select calls, avg_us from (select name, calls, `AverageNs`/1000.0 as avg_us from top_kernels) where name like '%k_scan_region%'") if False else None
            cur = sqlite3.connect(t[0]).cursor()
            us = None
            for name, calls, tot, avg, pct in cur.execute("select * from top_kernels"):
                if "k_scan_region" in name:
                    us = avg
            vals = {}
            for k, cn, n, avg in one(p[0], "select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%k_scan_region%' group by kernel_name, counter_name"):
                vals[cn] = avg
            rows[c] = (us, vals)
        if not rows:
            continue
        print(f"## scale group {g} ({'small, s <= 51' if g == 0 else 'mid, 51 < s <= 148'}), {frames} frames per launch, the group's launch alone on the chip")
        print("%-30s %10s | %s" % ("cut", "kernel us", " ".join("%18s" % c.replace("SQ_", "") for c in CTRS)))
        for c in CUTS:
            if c in rows:
                us, v = rows[c]
                print("%-30s %10.1f | %s" % (NAMES[c], us or -1, " ".join("%18.4g" % v.get(k, float('nan')) for k in CTRS)))
        print("-- what each stage adds (difference to the previous cut); cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs); LDS busy = IDX_ACTIVE / 256 CUs / cycles; "
              "VALU busy = INSTS_VALU x 2 cycles (wave64 on a SIMD-32) / 1024 SIMDs / cycles; conflict share = BANK_CONFLICT / IDX_ACTIVE")
        prev = None
        for c in CUTS:
            if c not in rows:
                continue
            us, v = rows[c]
            if prev is None:
                dus, dv = us, dict(v)
            else:
                dus = us - prev[0]
                dv = {k: v.get(k, 0) - prev[1].get(k, 0) for k in CTRS}
            ia, bc, il, iv = dv.get("SQ_LDS_IDX_ACTIVE", 0), dv.get("SQ_LDS_BANK_CONFLICT", 0), dv.get("SQ_INSTS_LDS", 0), dv.get("SQ_INSTS_VALU", 0)
            cyc = dv.get("GRBM_GUI_ACTIVE", 0) / 8.0
            print("%-30s %9.1f us (%4.1f %%) | LDS insts %.4g  LDS cycles/inst %.2f  conflict share %.2f  LDS busy %.2f | VALU insts %.4g  VALU busy %.2f | per CU: LDS cycles %.4g + VALU cycles %.4g = %.4g against %.4g measured" %
                  (NAMES[c] if prev is None else NAMES[c].lstrip("+ "), dus, 100.0 * dus / rows[0][0] if 0 in rows else 0, il, ia / max(il, 1), bc / max(ia, 1), ia / 256 / max(cyc, 1),
                   iv, iv * 2 / 1024 / max(cyc, 1), ia / 256, iv * 2 / 1024, ia / 256 + iv * 2 / 1024, cyc))
            prev = (us, v)
        print()


if __name__ == "__main__":
    main()
