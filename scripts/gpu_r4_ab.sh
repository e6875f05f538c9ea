# A/B loop: bench lines (no side legs) under environment settings:  bash scripts/gpu_r4_ab.sh "NAME:VAR=V VAR2=V2" ...
# PYTEST_K="expr" runs that part of the GPU parity suite first (PYTEST_ALL=1: all of it); BENCH_ARGS adds bench.py arguments.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PIGO_TUNING=1  # the settings below are tuning switches (ignored without it)
mkdir -p gpurun_out/r4
if [ -n "$PYTEST_ALL" ]; then timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r4/pytest_all.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r4/pytest_all.log | cut -c1-300; fi
if [ -n "$PYTEST_K" ]; then timeout 600 python -m pytest tests -m gpu -q -x -k "$PYTEST_K" > gpurun_out/r4/pytest_ab.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r4/pytest_ab.log | cut -c1-300; fi
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gray --shard-frames 0 --no-single-frame ${BENCH_ARGS:-}"
for spec in "$@"; do
  name="${spec%%:*}"; envs="${spec#*:}"
  timeout 300 env $envs $B 2>gpurun_out/r4/ab_$name.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'], d.get('overlap_ms'), d['kernel_ms'], d['cluster_ms'], d.get('verified_frames'))" || tail -3 gpurun_out/r4/ab_$name.err
done
