# Round 5, call 16: the 1,024-frame shard leg fell from 43 to 49 ms inside bench.py -- alone in a process? which leg in front of it matters?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5q; mkdir -p $O
export PIGO_TUNING=1
timeout 300 python scripts/ab.py --frames 1024 --steps 3 --reps 2 --kernel-times "base1024:" 2>$O/ab.err | tail -1 | tee $O/ab.txt
unset PIGO_TUNING
pyleg() { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], "shard", (d.get("config3_shard") or {}).get("ms_per_step"), (d.get("config3_shard") or {}).get("mwindows_per_s"), "one", (d.get("single_frame") or {}).get("hbm_resident_ms"))
PY
}
timeout 400 python bench.py --no-cpu-baseline --no-gray --no-config-legs --verify-frames 0 --no-single-frame > $O/b_shard_only.json 2>$O/b1.err; pyleg $O/b_shard_only.json
timeout 400 python bench.py --no-cpu-baseline --no-config-legs --verify-frames 0 --no-single-frame > $O/b_shard_gray.json 2>$O/b2.err; pyleg $O/b_shard_gray.json
timeout 400 python bench.py --no-cpu-baseline --no-config-legs --verify-frames 0 > $O/b_shard_single.json 2>$O/b3.err; pyleg $O/b_shard_single.json
timeout 400 python bench.py --no-cpu-baseline --verify-frames 0 > $O/b_all.json 2>$O/b4.err; pyleg $O/b_all.json
