cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name="$1"; shift; env "$@" python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-gray $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'])"; }
EXTRA="--frames 128"
run f128_prio1 PIGO_TAIL_PRIO=1
run f128_prio0 PIGO_TAIL_PRIO=0
run f128_prio1_c8 PIGO_TAIL_PRIO=1 PIGO_PIPE_CHUNKS=8
run f128_prio1_c16 PIGO_TAIL_PRIO=1 PIGO_PIPE_CHUNKS=16
EXTRA="--frames 64"
run f64_prio1 PIGO_TAIL_PRIO=1
run f64_prio0 PIGO_TAIL_PRIO=0
run f64_prio1_c4 PIGO_TAIL_PRIO=1 PIGO_PIPE_CHUNKS=4
run f64_prio1_c8 PIGO_TAIL_PRIO=1 PIGO_PIPE_CHUNKS=8
