#!/usr/bin/env python3
"""profiles/rNN_traffic.json from the FETCH_SIZE / WRITE_SIZE (and, optionally, TCC_HIT / TCC_MISS) passes of the round's final script.

    python scripts/make_traffic.py <pmc_fetch.db> <pmc_write.db> <steps in the profiled run> <frames per step> [<pmc_tcc.db> [<commit> [<pmc_ea.db>]]]

Fabric-side (L2-miss) bytes per frame of the scan kernels = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 / frames: FETCH_SIZE is
doubled per the gfx950 note of MI355X_MICROARCH.md; on this path's byte gathers 2 x FETCH_SIZE equals TCC_MISS_sum x 128 B
(profiles/r03_pmc_l2.txt), so the factor holds here.  The figure includes Infinity-Cache hits: it bounds the HBM bytes from above.
"""
import json
import sqlite3
import sys


def per_step(db, counter, steps):
    cur = sqlite3.connect(db).cursor()
    out = {}
    q = "select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? group by kernel_name"
    for name, n, tot in cur.execute(q, (counter,)):
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        out[short] = {"dispatches_per_step": n / steps, "kib_per_step": tot / steps}
    return out


def main():
    fetch_db, write_db, steps, frames = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    tcc_db = sys.argv[5] if len(sys.argv) > 5 else None
    commit = sys.argv[6] if len(sys.argv) > 6 else None
    f, w = per_step(fetch_db, "FETCH_SIZE", steps), per_step(write_db, "WRITE_SIZE", steps)
    SCAN = ("k_scan", "k_tail", "k_big")  # (k_big_pool was missing from rounds 3-4's sums: +~1 MB per frame)
    scan = lambda d: sum(v["kib_per_step"] for k, v in d.items() if k.startswith(SCAN))  # noqa: E731
    fk, wk = scan(f), scan(w)
    rec = {
        "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `bench.py --frames %d --steps 5 --warmup 2`; "
                  "scan kernels only (k_scan_region, k_scan_big, k_big_pool, k_scan_tile*, k_tail_deep*)" % frames,
        "commit": commit,
        "frames_per_step": frames, "steps_in_profiled_run": steps,
        "fetch_kib_per_step": round(fk, 1), "write_kib_per_step": round(wk, 1),
        "correction": "FETCH_SIZE doubled (gfx950 rocprofv3 note; upper bound for 4 B/lane copies), WRITE_SIZE as reported",
        "fabric_bytes_per_frame": int((2 * fk + wk) * 1024 / frames),
        "fabric_bytes_per_frame_uncorrected": int((fk + wk) * 1024 / frames),
        "note": "L2-miss traffic as seen on the fabric side of the L2s; Infinity-Cache hits are included, so this bounds the HBM bytes from above",
        "breakdown_fetch_kib_per_step": {k: round(v["kib_per_step"], 1) for k, v in sorted(f.items()) if k.startswith(("k_scan", "k_tail", "k_big", "k_restore", "k_sort", "k_cluster"))},
        "breakdown_write_kib_per_step": {k: round(v["kib_per_step"], 1) for k, v in sorted(w.items()) if k.startswith(("k_scan", "k_tail", "k_big", "k_restore", "k_sort", "k_cluster"))},
    }
    if tcc_db:
        cur = sqlite3.connect(tcc_db).cursor()
        l2 = {}
        q = "select kernel_name, counter_name, sum(value) from counters_collection where counter_name in ('TCC_HIT_sum', 'TCC_MISS_sum') group by kernel_name, counter_name"
        for name, cname, tot in cur.execute(q):
            short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            if short.startswith(("k_scan", "k_tail", "k_big")):
                l2.setdefault(short, {})[cname] = tot / steps
        rec["l2_per_step"] = {k: {"hits": round(v.get("TCC_HIT_sum", 0)), "misses": round(v.get("TCC_MISS_sum", 0)),
                                  "hit_rate": round(v.get("TCC_HIT_sum", 0) / max(1.0, v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0)), 4),
                                  "miss_bytes_128": round(v.get("TCC_MISS_sum", 0) * 128)} for k, v in sorted(l2.items())}
    ea_db = sys.argv[7] if len(sys.argv) > 7 else None
    if ea_db:
        # the L2s' memory-side read requests, all and "destined for DRAM (MC)": no counter of this stack separates Infinity-Cache
        # hits -- the cache sits on the memory side, behind this interface -- so DRAM-destined == everything that is not IO / GMI
        cur = sqlite3.connect(ea_db).cursor()
        tot = {}
        q = "select kernel_name, counter_name, sum(value) from counters_collection group by kernel_name, counter_name"
        for name, cname, v in cur.execute(q):
            short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            if short.startswith(("k_scan", "k_tail", "k_big")):
                tot[cname] = tot.get(cname, 0.0) + v / steps
        rd, rd32, dram = tot.get("TCC_EA0_RDREQ_sum", 0.0), tot.get("TCC_EA0_RDREQ_32B_sum", 0.0), tot.get("TCC_EA0_RDREQ_DRAM_sum", 0.0)
        rec["ea_read_requests_per_step"] = {k: round(v) for k, v in sorted(tot.items())}
        rec["ea_read_bytes_per_frame_64B_requests"] = int(((rd - rd32) * 64 + rd32 * 32) / frames)
        rec["dram_destined_share_of_read_requests"] = round(dram / max(rd, 1.0), 4)
        rec["hbm_side_note"] = ("TCC_EA0_RDREQ_DRAM counts the L2 read requests destined for DRAM (MC) as opposed to IO / GMI; the Infinity Cache is a "
                                "memory-side cache BEHIND that interface, so this stack has no counter for what it absorbs: the fabric-side bytes remain an "
                                "upper bound of the HBM bytes.  With %d frames per step (%.2f GB resident, the cache holds 0.27) whatever is re-read more "
                                "than ~100 frames later cannot come from it." % (frames, frames * 2.0736e-3))
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
