# Round-2 check: GPU parity suite, smoke, the default bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r2/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r2/pytest_gpu.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r2/smoke.log
timeout 900 python bench.py > gpurun_out/r2/bench_default.json 2> gpurun_out/r2/bench_default.err; echo "bench rc=$?"; cat gpurun_out/r2/bench_default.json; tail -5 gpurun_out/r2/bench_default.err
timeout 300 python bench.py --force-dist --frames 32 --steps 3 --warmup 1 --no-cpu-baseline --no-gray > gpurun_out/r2/bench_dist1.json 2> gpurun_out/r2/bench_dist1.err; echo "dist1 rc=$?"; cut -c1-600 gpurun_out/r2/bench_dist1.json; tail -3 gpurun_out/r2/bench_dist1.err
