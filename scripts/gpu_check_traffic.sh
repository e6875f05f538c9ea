cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_t
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python bench.py --frames 64 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['frames_per_s'], d['kernel_ms'])"
timeout 600 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof_t/f -o p -- python bench.py --frames 64 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_t/f.log 2>&1; echo "pmc rc=$?"
