# A/B on two configurations (default batch and the 4K config): bash scripts/gpu_r4_ab2.sh "NAME:ENV=.. ENV2=.." ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PIGO_TUNING=1  # the settings below are tuning switches (ignored without it)
O=gpurun_out/r4; mkdir -p $O
C="--no-cpu-baseline --no-gray --shard-frames 0 --no-single-frame"
show() { python -c "import json,sys; d=json.loads(open('$1').read()); print('$2', d['value'], d['ms_per_step'], d.get('overlap_ms'), d['kernel_ms'], d['cluster_ms'], d.get('verified_frames'))" || tail -3 ${1%.json}.err; }
for spec in "$@"; do
  name="${spec%%:*}"; envs="${spec#*:}"
  timeout 300 env $envs python bench.py $C ${BENCH_ARGS:-} > $O/ab2_${name}_faces.json 2> $O/ab2_${name}_faces.err; show $O/ab2_${name}_faces.json "$name faces"
  if [ -z "$NO4K" ]; then timeout 300 env $envs python bench.py --rows 2160 --cols 3840 --min-size 20 --max-size 2000 --shift 0.05 --scale 1.05 --frames 8 --det-cap 32768 --gather-cap 64 --steps 5 --warmup 2 --verify-frames 2 $C > $O/ab2_${name}_4k.json 2> $O/ab2_${name}_4k.err; show $O/ab2_${name}_4k.json "$name 4k"; fi
done
