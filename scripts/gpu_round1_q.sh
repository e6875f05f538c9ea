cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_q
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | cut -c1-200
run() { name="$1"; shift; env "$@" python bench.py --frames 64 --steps 5 --warmup 2 --no-cpu-baseline $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['frames_per_s'], d['kernel_ms'], d['config']['head_survivor_fraction'])"; }
run base X=1
EXTRA="--kind noise" run noise X=1
EXTRA="--angle 0.8" run rot X=1
EXTRA="--frames 13" run f13 X=1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof_q/pmc_fetch -o p -- python bench.py --frames 64 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_q/pmc_fetch.log 2>&1; echo "pmc rc=$?"
python scripts/single_frame_latency.py 2>&1 | grep "single 1080p"
