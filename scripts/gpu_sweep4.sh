run() { name="$1"; shift; env "$@" python bench.py --frames 64 --steps 10 --warmup 3 --no-cpu-baseline $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['frames_per_s'], 'ms/step', d['ms_per_step'], d['kernel_ms'])"; }
run side X=1
run noside PIGO_SIDE_STREAM=0
EXTRA="--kind noise" run noise_side X=1
EXTRA="--kind noise" run noise_noside PIGO_SIDE_STREAM=0
EXTRA="--frames 128" run side128 X=1
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
