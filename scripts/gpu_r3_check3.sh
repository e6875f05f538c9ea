cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
bash scripts/gpu_r3_stress.sh
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_parity.py::test_slot_capture_next_to_plan_builds_on_other_handles -k "${1:-patch or benchmarked or 4k or region_deep or batch_api or golden}" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest_gpu.log | cut -c1-300
run() { name="$1"; shift; timeout 300 env "$@" > $O/bench_$name.json 2>$O/bench_$name.err; python -c "
import json,sys
try:
    d=json.load(open('$O/bench_$name.json')); print('$name', d['value'], d['ms_per_step'], d['kernel_ms'], 'cluster', d['cluster_ms'], 'verified', d.get('verified_frames'), d.get('gather'))
except Exception as e:
    print('$name FAILED', e); print(open('$O/bench_$name.err').read()[-1500:])
"; }
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gray --shard-frames 0"
run default X=1 $B
run nopatch PIGO_PATCH=0 $B
run k4 X=1 python bench.py --rows 2160 --cols 3840 --min-size 20 --max-size 2000 --shift 0.05 --scale 1.05 --frames 8 --det-cap 32768 --gather-cap 64 --steps 5 --warmup 2 --no-cpu-baseline --no-gray --shard-frames 0 --verify-frames 1
