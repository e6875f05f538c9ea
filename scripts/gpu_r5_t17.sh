# Round 5, call 17: ClusterDetections of a short host list as ONE launch on pinned host memory; shard leg with three warm-up steps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5r; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -m gpu -x -q -k "golden or cluster or Cluster or tie or sweep or reentrant or pipeline or mirror" 2>&1 | tail -5 | tee $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --no-cpu-baseline --verify-frames 8 > $O/bench.json 2>$O/bench.err || tail -5 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5r/bench.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d.get("single_frame"), {k:d["reference_benchmark"][k] for k in ("gpu_ms_per_op","gpu_scan_only_ms_per_op")}, d["config3_shard"]["ms_per_step"], d["config3_shard"]["mwindows_per_s"])
PY
