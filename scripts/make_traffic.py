#!/usr/bin/env python3
"""profiles/r01_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of scripts/gpu_round1_final.sh.

    python scripts/make_traffic.py <pmc_fetch.db> <pmc_write.db> <steps in the profiled run> <frames per step>

HBM bytes per frame of the scan kernels (k_scan_tile*, k_tail_deep*) = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 / frames:
FETCH_SIZE is doubled per the gfx950 note of MI355X_MICROARCH.md (rocprofv3 reports half of a wide coalesced stream; for
this kernel's 4-byte-per-lane copies that is an upper bound), WRITE_SIZE is taken as reported.
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
    f, w = per_step(fetch_db, "FETCH_SIZE", steps), per_step(write_db, "WRITE_SIZE", steps)
    scan = lambda d: sum(v["kib_per_step"] for k, v in d.items() if k.startswith("k_scan") or k.startswith("k_tail"))  # noqa: E731
    fk, wk = scan(f), scan(w)
    rec = {
        "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `bench.py --frames %d --steps 5 --warmup 2`; "
                  "scan kernels only (k_scan_region, k_scan_tile*, k_tail_deep*)" % frames,
        "frames_per_step": frames, "steps_in_profiled_run": steps,
        "fetch_kib_per_step": round(fk, 1), "write_kib_per_step": round(wk, 1),
        "correction": "FETCH_SIZE doubled (gfx950 rocprofv3 note; upper bound for 4 B/lane copies), WRITE_SIZE as reported",
        "hbm_bytes_per_frame": int((2 * fk + wk) * 1024 / frames),
        "hbm_bytes_per_frame_uncorrected": int((fk + wk) * 1024 / frames),
        "breakdown_fetch_kib_per_step": {k: round(v["kib_per_step"], 1) for k, v in sorted(f.items()) if k.startswith(("k_scan", "k_tail", "k_restore", "k_sort", "k_cluster"))},
        "breakdown_write_kib_per_step": {k: round(v["kib_per_step"], 1) for k, v in sorted(w.items()) if k.startswith(("k_scan", "k_tail", "k_restore", "k_sort", "k_cluster"))},
    }
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
