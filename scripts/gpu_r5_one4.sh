# Round 5, call 4: k_scan_one restores the order itself (one stream node per call), late entries walk four trees per lane; deep-list caps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5d; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest.txt
export PIGO_TUNING=1
S=("one:" "v2:PIGO_SCAN_VARIANT=2" "norestore:PIGO_ONE_RESTORE=0" "late0:PIGO_ONE_LATE_ITEMS=0" "late32:PIGO_ONE_LATE_ITEMS=32" "late96:PIGO_ONE_LATE_ITEMS=96" "ntl2:PIGO_ONE_NT_LATE=2"
   "d64:PIGO_ONE_DEEP0=64 PIGO_ONE_DEEP1=64" "d128:PIGO_ONE_DEEP0=128 PIGO_ONE_DEEP1=128" "d256:PIGO_ONE_DEEP0=256 PIGO_ONE_DEEP1=256"
   "noquad:PIGO_REG_QUAD0=0 PIGO_REG_QUAD1=0" "mid28:PIGO_NH_REG1=28" "w20:PIGO_ONE_W1_X10=20" "w45:PIGO_ONE_W1_X10=45" "s240:PIGO_ONE_SLOTS=240" "local2:PIGO_ONE_LOCAL0=2 PIGO_ONE_LOCAL1=2" "one_b:")
timeout 400 python scripts/ab.py --frames 1 --steps 100 --no-cluster --kernel-times "${S[@]}" 2>$O/ab_one.err | tee $O/ab_one.txt || tail -5 $O/ab_one.err
timeout 200 python scripts/ab.py --frames 1 --steps 100 --no-cluster --kind noise "one:" "v2:PIGO_SCAN_VARIANT=2" 2>$O/ab_one_noise.err | tee $O/ab_one_noise.txt || tail -5 $O/ab_one_noise.err
timeout 200 python scripts/ab.py --frames 1 --steps 100 --no-cluster --angle 0.8 "one:" "v2:PIGO_SCAN_VARIANT=2" 2>$O/ab_one_rot.err | tee $O/ab_one_rot.txt || tail -5 $O/ab_one_rot.err
timeout 200 python scripts/ab.py --frames 3 --steps 50 --no-cluster "one:" "v2:PIGO_SCAN_VARIANT=2" 2>$O/ab_three.err | tee $O/ab_three.txt || tail -5 $O/ab_three.err
export PIGO_HIP_LIB=$GRAFT_REPO_ROOT/pigo_amd/csrc/libpigo_hip_debug.so
timeout 120 env PIGO_SYNC_DEBUG=1 python scripts/one_trace.py 2>$O/trace.err | tee $O/trace.txt || tail -5 $O/trace.err
grep "k_scan_one" $O/trace.err | head -2
timeout 120 python scripts/one_trace.py --rows 400 --cols 320 --shift 0.2 2>>$O/trace.err | tee -a $O/trace.txt
unset PIGO_HIP_LIB PIGO_TUNING
timeout 200 python scripts/single_frame_latency.py 2>&1 | tail -2 | tee $O/single.txt
timeout 400 python bench.py --no-cpu-baseline --shard-frames 0 --verify-frames 8 > $O/bench.json 2> $O/bench.err || tail -5 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5d/bench.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d.get("single_frame"), {k:d["reference_benchmark"][k] for k in ("gpu_ms_per_op","gpu_scan_only_ms_per_op")}, {k:v.get("ms_per_step") for k,v in d.items() if isinstance(v,dict) and "ms_per_step" in v})
PY
