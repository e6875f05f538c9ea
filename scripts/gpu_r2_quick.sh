# quick loop: GPU parity suite (stop at first failure) + a short bench line without the side legs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 1500 python -m pytest tests -m gpu -q -x ${PYTEST_ARGS:-} > gpurun_out/r2/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r2/pytest_gpu.log | cut -c1-250
run() { name="$1"; shift; env "$@" 2>gpurun_out/r2/$name.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'], d['kernel_ms'], d['cluster_ms'], d.get('verified_frames'))" || tail -5 gpurun_out/r2/$name.err; grep debug_stats gpurun_out/r2/$name.err; }
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gray --shard-frames 0"
run default X=1 $B
run rot X=1 $B --angle 0.8
run noise X=1 $B --kind noise
run dbg64 PIGO_HIP_LIB=$GRAFT_REPO_ROOT/pigo_amd/csrc/libpigo_hip_debug.so PIGO_DEBUG_STATS=1 $B --frames 64 --verify-frames 0
