# Round-2 diagnostics: where do the cycles of a tile go, per class of scales (PIGO_DEBUG_STATS phase timers)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
run() { name="$1"; shift; env PIGO_DEBUG_STATS=1 "$@" 2>gpurun_out/r2/$name.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'], d['kernel_ms'], d['config']['windows_per_frame'], d['config']['head_survivor_fraction'])"; grep debug_stats gpurun_out/r2/$name.err; }
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-gray --frames 64"
run all X=1 $B
run all_nochunk PIGO_PIPE_CHUNKS=1 PIGO_SIDE_STREAM=0 $B
run small_20_28 X=1 $B --min-size 20 --max-size 29
run small_30_46 X=1 $B --min-size 30 --max-size 48
run mid_glb X=1 $B --min-size 50 --max-size 135
run mid_l5_16_40k PIGO_TILE_RULES="5,16,40960" $B --min-size 50 --max-size 135
run mid_l5_16_80k PIGO_TILE_RULES="5,16,81920" $B --min-size 50 --max-size 135
run mid_l5_16_150k PIGO_TILE_RULES="5,16,153600" $B --min-size 50 --max-size 135
run mid_l6_16_80k PIGO_TILE_RULES="6,16,81920" $B --min-size 50 --max-size 135
run mid_l6_16_150k PIGO_TILE_RULES="6,16,153600" $B --min-size 50 --max-size 135
run mid_l6_8_80k PIGO_TILE_RULES="6,8,81920" $B --min-size 50 --max-size 135
run big_glb X=1 $B --min-size 140 --max-size 1000
