# validation of a kernel change: the tests named in $1 (a -k expression) + default / 4K bench lines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 1200 python -m pytest tests -m gpu -q -x -k "$1" > gpurun_out/r2/pytest_val.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r2/pytest_val.log | cut -c1-250
run() { name="$1"; shift; env "$@" 2>gpurun_out/r2/$name.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'], d['kernel_ms'], d['cluster_ms'], d.get('verified_frames'))" || tail -5 gpurun_out/r2/$name.err; }
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gray --shard-frames 0"
run default X=1 $B
run k4 X=1 python bench.py --rows 2160 --cols 3840 --min-size 20 --max-size 2000 --shift 0.05 --scale 1.05 --frames 8 --det-cap 32768 --gather-cap 64 --steps 5 --warmup 2 --no-cpu-baseline --no-gray --shard-frames 0 --verify-frames 1
