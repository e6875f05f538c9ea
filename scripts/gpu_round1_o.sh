mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/pytest_gpu.log | cut -c1-200
run() { name=$1; shift; timeout 300 python bench.py --frames 64 --steps 5 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/bench_$name.json 2> gpurun_out/bench_$name.err; echo "bench $name rc=$?"; grep debug_stats gpurun_out/bench_$name.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/bench_$name.json")); print("$name", d["value"], "Mwin/s", d["frames_per_s"], "fps", d["kernel_ms"], "surv", d["config"]["head_survivor_fraction"])
except Exception as e: print("$name FAILED", e); print(open("gpurun_out/bench_$name.err").read()[-1500:])
PY
}
run default
PIGO_DEBUG_STATS=1 run dbg_default
PIGO_NH_LDS=18 run l18
PIGO_NH_LDS=18 PIGO_NH_GLB=8 run l18_g8
run noise --kind noise
run rot --angle 0.8
