mkdir -p gpurun_out
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpuinfo.txt; nproc >> gpurun_out/gpuinfo.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" 
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -30 gpurun_out/pytest_gpu.log
for v in 1 0; do for k in faces noise; do
timeout 300 python bench.py --variant $v --kind $k --frames 64 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_v${v}_${k}.json 2> gpurun_out/bench_v${v}_${k}.err; echo "bench v$v $k rc=$?"; cat gpurun_out/bench_v${v}_${k}.json; tail -3 gpurun_out/bench_v${v}_${k}.err
done; done
