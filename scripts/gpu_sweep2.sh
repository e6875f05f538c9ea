cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name="$1"; shift; env "$@" python bench.py --frames 64 --steps 8 --warmup 2 --no-cpu-baseline --no-gray 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['frames_per_s'], d['kernel_ms'])"; }
run base X=1
run mid32x16_40k PIGO_TILE_RULES="6,32,16384;6,16,40960;5,16,40960"
run mid32x16_48k PIGO_TILE_RULES="6,32,16384;6,16,40960;5,16,49152"
run mid32x16_64k PIGO_TILE_RULES="6,32,16384;6,16,40960;5,16,65536"
run a32_15k PIGO_TILE_RULES="6,32,15360;6,16,40960"
run a32_14k_mid PIGO_TILE_RULES="6,32,14336;6,16,40960;5,16,40960"
run base2 X=1
