mkdir -p gpurun_out
run() { name=$1; shift; timeout 300 python bench.py --frames 64 --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/bench_$name.json 2> gpurun_out/bench_$name.err; echo "bench $name rc=$?"; grep debug_stats gpurun_out/bench_$name.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/bench_$name.json")); print("$name", d["value"], "Mwin/s", d["frames_per_s"], "fps", d["kernel_ms"], "surv", d["config"]["head_survivor_fraction"])
except Exception as e: print("$name FAILED", e); print(open("gpurun_out/bench_$name.err").read()[-1500:])
PY
}
# classes are launched in the order of their first scale, so with these rules the FIRST launch (blocks < 65536) is the small-scale LDS class
PIGO_DEBUG_STATS=1 PIGO_TILE_RULES="6,32,16384" run dbg_small32
PIGO_DEBUG_STATS=1 PIGO_TILE_RULES="6,16,16384" run dbg_small16
PIGO_DEBUG_STATS=1 PIGO_TILE_RULES="6,8,16384" run dbg_small8
PIGO_DEBUG_STATS=1 PIGO_LDS_TILES=0 run dbg_glb
