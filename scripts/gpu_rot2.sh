cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3 | cut -c1-300
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | cut -c1-300
run() { name="$1"; shift; env "$@" python bench.py --frames 64 --steps 8 --warmup 2 --no-cpu-baseline --no-gray --angle 0.8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['frames_per_s'], d['kernel_ms'], d['config']['head_survivor_fraction'])"; }
run rot_lds X=1
run rot_lds_qb2 PIGO_ROT_QB_DIV=2
env X=1 python bench.py --frames 64 --steps 8 --warmup 2 --no-cpu-baseline --no-gray --angle 0.2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rot_a02', d['value'], d['kernel_ms'])"
