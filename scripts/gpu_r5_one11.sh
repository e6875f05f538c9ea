# Round 5, call 11: probe deleted, results of RunCascade written to pinned host memory by the launch, new tests, bench legs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5k; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 | tee $O/pytest.txt
timeout 200 python scripts/single_frame_latency.py 2>&1 | tail -2 | tee $O/single.txt
timeout 600 python bench.py --no-cpu-baseline --shard-frames 0 --verify-frames 8 > $O/bench.json 2> $O/bench.err || tail -5 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5k/bench.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d.get("single_frame"), {k:d["reference_benchmark"][k] for k in ("gpu_ms_per_op","gpu_scan_only_ms_per_op")}, {k:v.get("ms_per_step") for k,v in d.items() if isinstance(v,dict) and "ms_per_step" in v}, d["roofline"]["frac"])
PY
