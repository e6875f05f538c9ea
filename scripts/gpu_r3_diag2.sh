# Round-3 diagnostics 2: cache-policy microbenchmark, kernel traces of the 4K config and the default config
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
[ -x scripts/micro/ta_policy ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/ta_policy.hip -o scripts/micro/ta_policy
timeout 120 scripts/micro/ta_policy 2000 > $O/ta_policy.txt 2>&1; echo "ta_policy rc=$?"; grep "waves/CU 16" $O/ta_policy.txt | cut -c1-150
K4="python bench.py --rows 2160 --cols 3840 --min-size 20 --max-size 2000 --shift 0.05 --scale 1.05 --frames 8 --det-cap 32768 --gather-cap 64 --steps 3 --warmup 1 --no-cpu-baseline --no-gray --shard-frames 0 --verify-frames 0 --no-single-frame"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace4k -o t -- $K4 > $O/trace4k.log 2>&1; echo "trace4k rc=$?"
python scripts/summarize_prof.py "4K config, 8 frames per step: $K4" $O/trace4k/t_results.db > $O/trace4k_summary.txt 2>&1; head -30 $O/trace4k_summary.txt | cut -c1-160
B="env PIGO_SIDE_STREAM=0 python bench.py --frames 64 --steps 5 --warmup 2 --no-cpu-baseline --no-single-frame --shard-frames 0 --verify-frames 0 --no-gray"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- $B > $O/trace.log 2>&1; echo "trace rc=$?"
python scripts/summarize_prof.py "default config, 64 frames per step: $B" $O/trace/t_results.db > $O/trace_summary.txt 2>&1; head -30 $O/trace_summary.txt | cut -c1-160
rm -rf $O/trace $O/trace4k
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gray --shard-frames 0 2>$O/bench_default.err | tee $O/bench_default.json | cut -c1-300
