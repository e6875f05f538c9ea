cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "rc=$?"; tail -2 gpurun_out/pytest_gpu.log | cut -c1-300; grep -n "^E  " gpurun_out/pytest_gpu.log | head -6
python bench.py --frames 128 --steps 6 --warmup 2 --no-cpu-baseline --no-gray 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['cluster_ms'])"
