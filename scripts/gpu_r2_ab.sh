# A/B loop: bench lines (no side legs) under a list of environment settings:  bash scripts/gpu_r2_ab.sh "NAME:VAR=V VAR2=V2" ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gray --shard-frames 0 ${BENCH_ARGS:-}"
for spec in "$@"; do
  name="${spec%%:*}"; envs="${spec#*:}"
  env $envs $B 2>gpurun_out/r2/ab_$name.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'], d['kernel_ms'], d['cluster_ms'], d.get('verified_frames'))" || tail -3 gpurun_out/r2/ab_$name.err
  grep debug_stats gpurun_out/r2/ab_$name.err
done
