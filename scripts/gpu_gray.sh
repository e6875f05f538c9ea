# f1 (RgbToGrayscale) bring-up: GPU parity suite + bench line with the gray side measurement
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | cut -c1-400
timeout 300 python bench.py --frames 64 --steps 10 --warmup 3 --no-cpu-baseline 2>gpurun_out/bench_gray.err | tee gpurun_out/bench_gray.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['frames_per_s'], d['kernel_ms']); print(d['gray'])"
tail -3 gpurun_out/bench_gray.err
