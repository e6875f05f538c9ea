run() { name="$1"; shift; env "$@" python bench.py --frames 64 --steps 5 --warmup 2 --no-cpu-baseline $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['frames_per_s'], d['kernel_ms'])"; }
run base X=1
run qb3 PIGO_QB_EIGHTHS=3
run u8 PIGO_U8=1
run qb3_u8 PIGO_QB_EIGHTHS=3 PIGO_U8=1
EXTRA="--kind noise" run noise_qb3 PIGO_QB_EIGHTHS=3
EXTRA="--kind noise" run noise_base X=1
PIGO_QB_EIGHTHS=3 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
