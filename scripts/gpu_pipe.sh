cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -4 | cut -c1-400
run() { name="$1"; shift; env "$@" python bench.py --frames 64 --steps 8 --warmup 2 --no-cpu-baseline --no-gray $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'])"; }
run c1 PIGO_PIPE_CHUNKS=1
run c2 PIGO_PIPE_CHUNKS=2
run c4 PIGO_PIPE_CHUNKS=4
run c8 PIGO_PIPE_CHUNKS=8
EXTRA="--frames 128"
run f128_c1 PIGO_PIPE_CHUNKS=1
run f128_c2 PIGO_PIPE_CHUNKS=2
run f128_c4 PIGO_PIPE_CHUNKS=4
run f128_c8 PIGO_PIPE_CHUNKS=8
