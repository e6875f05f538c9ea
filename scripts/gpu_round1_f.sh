mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/pytest_gpu.log
run() { name=$1; shift; timeout 300 python bench.py --frames 64 --steps 5 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/bench_$name.json 2> gpurun_out/bench_$name.err; echo "bench $name rc=$?"; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/bench_$name.json")); print("$name", d["value"], "Mwin/s", d["frames_per_s"], "fps", d["kernel_ms"], "cluster", d["cluster_ms"], "surv", d["config"]["head_survivor_fraction"])
except Exception as e: print("$name FAILED", e); print(open("gpurun_out/bench_$name.err").read()[-1500:])
PY
}
run v2_default
run v2_noise --kind noise
PIGO_TILE_RULES="6,32,16384;6,16,24576;6,8,36864" run v2_nolarge
PIGO_TILE_RULES="6,32,16384;6,16,24576;6,8,36864" PIGO_NH_GLB=8 run v2_nolarge_g8
PIGO_TILE_RULES="6,32,16384;6,16,24576;6,8,36864" PIGO_NH_LDS=28 run v2_nolarge_l28
PIGO_TILE_RULES="6,16,24576;6,8,36864" run v2_t16_nolarge
PIGO_TILE_RULES="6,32,16384" run v2_onlysmall
PIGO_LDS_TILES=0 run v2_global
run v2_rot --angle 0.8
