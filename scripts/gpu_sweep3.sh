cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name="$1"; shift; env "$@" python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-gray --frames 128 $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'], d['kernel_ms'])"; }
run base X=1
run r20k PIGO_TILE_RULES="6,32,20480;6,16,40960"
run r12k PIGO_TILE_RULES="6,32,12288;6,16,40960"
run r16k_48k PIGO_TILE_RULES="6,32,16384;6,16,49152"
run r16k_32k PIGO_TILE_RULES="6,32,16384;6,16,32768"
run nh24 PIGO_NH_LDS=24 PIGO_NH_GLB=24
run nh32 PIGO_NH_LDS=32 PIGO_NH_GLB=32
run nh28_g18 PIGO_NH_GLB=18
run gth8 PIGO_GLOBAL_TH=8
run gth32 PIGO_GLOBAL_TH=32
