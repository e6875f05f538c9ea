# bench.py's default line (as the driver runs it) + the single-frame script (+ PYTEST_K: a part of the GPU parity suite)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; mkdir -p $O
if [ -n "$PYTEST_ALL" ]; then timeout 900 python -m pytest tests -m gpu -q > $O/pytest_final.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_final.log | tail -1 | tee $O/pytest_final_tail.txt; fi
if [ -n "$PYTEST_K" ]; then timeout 600 python -m pytest tests -m gpu -q -x -k "$PYTEST_K" 2>&1 | tail -2; fi
timeout 600 python bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"
python -c "
import json;d=json.load(open('$O/bench_final.json'))
print(d['value'], d['ms_per_step'], d['kernel_ms'], 'cluster', d['cluster_ms'], 'overlap', d['overlap_ms'], 'shard', d['config3_shard']['mwindows_per_s'], 'rot', d['config4_rotated']['upright_faces']['mwindows_per_s'], d['config4_rotated']['rotated_faces']['mwindows_per_s'], '4k', d['config5_4k']['mwindows_per_s'], 'single', d['single_frame']['hbm_resident_ms'], d['single_frame']['host_buffer_ms'], 'ref', d['reference_benchmark']['gpu_ms_per_op'], 'cpu', d['cpu_baseline']['value'])"
python scripts/single_frame_latency.py 2>&1 | grep "single 1080p" | tee $O/single_frame_final.txt
