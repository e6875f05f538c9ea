run() { name="$1"; shift; env "$@" python bench.py --frames 64 --steps 10 --warmup 3 --no-cpu-baseline $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['frames_per_s'], 'ms/step', d['ms_per_step'])"; }
run side1 PIGO_SIDE_STREAM=1
run side2 PIGO_SIDE_STREAM=2
run side2_rules3 PIGO_SIDE_STREAM=2 PIGO_TILE_RULES="6,32,16384;6,16,24576;6,16,40960"
EXTRA="--kind noise" run noise_side2 PIGO_SIDE_STREAM=2
PIGO_SIDE_STREAM=2 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
