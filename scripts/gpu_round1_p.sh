timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | cut -c1-200
run() { name="$1"; shift; env "$@" python bench.py --frames 64 --steps 5 --warmup 2 --no-cpu-baseline $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['frames_per_s'], d['kernel_ms'])"; }
run base X=1
run split64 PIGO_DEEP_SPLIT=64
run split192 PIGO_DEEP_SPLIT=192
run split440 PIGO_DEEP_SPLIT=440
EXTRA="--kind noise" run noise X=1
EXTRA="--angle 0.8" run rot X=1
