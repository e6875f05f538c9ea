# Round-3 check: parity suite (optionally a -k subset in $1), then default / rotated / 4K / 1-frame bench lines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q ${1:+-k "$1"} > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest_gpu.log | cut -c1-300
run() { name="$1"; shift; timeout 300 env "$@" > $O/bench_$name.json 2>$O/bench_$name.err; python -c "
import json,sys
try:
    d=json.load(open('$O/bench_$name.json')); print('$name', d['value'], d['ms_per_step'], d['kernel_ms'], 'cluster', d['cluster_ms'], 'verified', d.get('verified_frames'), d.get('gather'))
except Exception as e:
    print('$name FAILED', e); print(open('$O/bench_$name.err').read()[-1500:])
"; }
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gray --shard-frames 0"
run default X=1 $B
run rot X=1 $B --angle 0.8
run k4 X=1 python bench.py --rows 2160 --cols 3840 --min-size 20 --max-size 2000 --shift 0.05 --scale 1.05 --frames 8 --det-cap 32768 --gather-cap 64 --steps 5 --warmup 2 --no-cpu-baseline --no-gray --shard-frames 0 --verify-frames 1
run dist1 X=1 python bench.py --force-dist --frames 32 --steps 3 --warmup 1 --no-cpu-baseline --no-gray --shard-frames 0
