# f2/f3 bring-up: GPU parity suite + bench line with the side measurements
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | cut -c1-600
timeout 300 python bench.py --frames 64 --steps 10 --warmup 3 2>gpurun_out/bench_pup.err | tee gpurun_out/bench_pup.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['frames_per_s'], d['kernel_ms']); print(d['gray']); print(d['puploc'])"
tail -3 gpurun_out/bench_pup.err
