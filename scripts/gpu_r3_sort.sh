# Round-3: the wave-parallel Go-order sort -- cluster/tie tests, then the 4K and the default bench lines with a kernel trace of the 4K one
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "${1:-tie or go_order or cluster or 4k or batch_api}" > $O/pytest_sort.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_sort.log | cut -c1-300
K4="python bench.py --rows 2160 --cols 3840 --min-size 20 --max-size 2000 --shift 0.05 --scale 1.05 --frames 8 --det-cap 32768 --gather-cap 64 --steps 3 --warmup 1 --no-cpu-baseline --no-gray --shard-frames 0 --verify-frames 1 --no-single-frame"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace4k -o t -- $K4 > $O/trace4k.log 2>&1; echo "trace4k rc=$?"; grep '"metric"' $O/trace4k.log | cut -c1-250
python scripts/summarize_prof.py "4K config, 8 frames per step: $K4" $O/trace4k/t_results.db > $O/trace4k_summary.txt 2>&1; head -14 $O/trace4k_summary.txt | cut -c1-160
rm -rf $O/trace4k
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gray --shard-frames 0 2>$O/bench_default.err | tee $O/bench_default.json | cut -c1-300; tail -3 $O/bench_default.err
