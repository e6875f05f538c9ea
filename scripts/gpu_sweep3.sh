run() { name="$1"; shift; env "$@" python bench.py --frames 64 --steps 5 --warmup 2 --no-cpu-baseline $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['frames_per_s'], d['kernel_ms'])"; }
run base X=1
run tabglobal PIGO_TAB_GLOBAL=1
run tabglobal_r32 PIGO_TAB_GLOBAL=1 PIGO_TILE_RULES="6,32,32768;6,16,40960"
EXTRA="--kind noise" run noise_tabglobal PIGO_TAB_GLOBAL=1
PIGO_TAB_GLOBAL=1 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
