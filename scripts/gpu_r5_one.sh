# Round 5, call 1: the persistent one-frame launch (k_scan_one).  GPU suite first (parity), then one 1080p frame per plan: the new
# default against variant 2 and a few item splits, per-kernel times; then the default bench line (the batch path must not have moved).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest.txt
export PIGO_TUNING=1
S=("one:" "v2:PIGO_SCAN_VARIANT=2" "one_w20:PIGO_ONE_W1_X10=20" "one_w45:PIGO_ONE_W1_X10=45" "one_s224:PIGO_ONE_SLOTS=224" "one_s320:PIGO_ONE_SLOTS=320" "one_s512:PIGO_ONE_SLOTS=512" "one_b:")
timeout 300 python scripts/ab.py --frames 1 --steps 100 --no-cluster --kernel-times "${S[@]}" 2>$O/ab_one.err | tee $O/ab_one.txt || tail -5 $O/ab_one.err
timeout 200 python scripts/ab.py --frames 1 --steps 100 --no-cluster --kind noise "one:" "v2:PIGO_SCAN_VARIANT=2" 2>$O/ab_one_noise.err | tee $O/ab_one_noise.txt || tail -5 $O/ab_one_noise.err
timeout 200 python scripts/ab.py --frames 1 --steps 100 --no-cluster --angle 0.8 "one:" "v2:PIGO_SCAN_VARIANT=2" 2>$O/ab_one_rot.err | tee $O/ab_one_rot.txt || tail -5 $O/ab_one_rot.err
timeout 200 python scripts/ab.py --frames 3 --steps 50 --no-cluster "one:" "v2:PIGO_SCAN_VARIANT=2" 2>$O/ab_three.err | tee $O/ab_three.txt || tail -5 $O/ab_three.err
unset PIGO_TUNING
timeout 200 python scripts/single_frame_latency.py 2>&1 | tail -2 | tee $O/single.txt
timeout 400 python bench.py --no-cpu-baseline --shard-frames 0 > $O/bench.json 2> $O/bench.err || tail -5 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5a/bench.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d.get("single_frame"), d.get("reference_benchmark"), {k:v.get("ms_per_step") for k,v in d.items() if isinstance(v,dict) and "ms_per_step" in v})
PY
