# the other bench configurations (noise, rotated, rotated faces, 4K) under environment settings: bash scripts/gpu_r4_configs.sh NAME "ENV=.."
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PIGO_TUNING=1  # the settings below are tuning switches (ignored without it)
O=gpurun_out/r4; mkdir -p $O
name="$1"; envs="$2"
C="--no-cpu-baseline --no-gray --shard-frames 0 --no-single-frame"
show() { python -c "import json,sys; d=json.loads(open('$1').read()); print('$2', d['value'], d['ms_per_step'], d.get('overlap_ms'), d['kernel_ms'], d['cluster_ms'], d.get('verified_frames'))" || tail -3 ${1%.json}.err; }
timeout 300 env $envs python bench.py $C > $O/cfg_${name}_faces.json 2> $O/cfg_${name}_faces.err; show $O/cfg_${name}_faces.json faces
timeout 300 env $envs python bench.py --kind noise $C > $O/cfg_${name}_noise.json 2> $O/cfg_${name}_noise.err; show $O/cfg_${name}_noise.json noise
timeout 300 env $envs python bench.py --angle 0.8 $C > $O/cfg_${name}_rot.json 2> $O/cfg_${name}_rot.err; show $O/cfg_${name}_rot.json rot
timeout 300 env $envs python bench.py --angle 0.8 --face-rotation -79 $C > $O/cfg_${name}_rotfaces.json 2> $O/cfg_${name}_rotfaces.err; show $O/cfg_${name}_rotfaces.json rotfaces
timeout 300 env $envs python bench.py --rows 2160 --cols 3840 --min-size 20 --max-size 2000 --shift 0.05 --scale 1.05 --frames 8 --det-cap 32768 --gather-cap 64 --steps 5 --warmup 2 --verify-frames 2 $C > $O/cfg_${name}_4k.json 2> $O/cfg_${name}_4k.err; show $O/cfg_${name}_4k.json 4k
