cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_parity.py::test_slot_capture_next_to_plan_builds_on_other_handles > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest_gpu.log | cut -c1-300
timeout 300 rocgdb -batch -ex "handle SIGABRT stop print" -ex "handle SIGSEGV stop print" -ex run -ex "thread apply all bt 30" --args python -m pytest tests/test_gpu_parity.py -q -x -k test_slot_capture -p no:cacheprovider > $O/abort2_gdb.txt 2>&1
echo "gdb rc=$? $(grep -c -E 'SIGABRT|SIGSEGV' $O/abort2_gdb.txt) signals; $(grep -E 'passed|failed' $O/abort2_gdb.txt | tail -1)"
grep -n -A45 "received signal" $O/abort2_gdb.txt | cut -c1-200 | head -120
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace4k -o t -- python bench.py --rows 2160 --cols 3840 --min-size 20 --max-size 2000 --shift 0.05 --scale 1.05 --frames 8 --det-cap 32768 --gather-cap 64 --steps 3 --warmup 1 --no-cpu-baseline --no-gray --shard-frames 0 --verify-frames 0 --no-single-frame > $O/trace4k.log 2>&1; echo "trace4k rc=$?"
python scripts/summarize_prof.py "4K config kernel trace" $O/trace4k/t_results.db | head -24; rm -rf $O/trace4k
