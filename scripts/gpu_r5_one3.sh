# Round 5, call 3: k_scan_one with the poison termination (no polled shared word)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "reentrant or golden or run_cascade or sweep or edge" 2>&1 | tail -8 | tee $O/pytest.txt
export PIGO_TUNING=1
S=("one:" "v2:PIGO_SCAN_VARIANT=2" "nt1:PIGO_ONE_NT=1" "nt2:PIGO_ONE_NT=2" "local0:PIGO_ONE_LOCAL0=0 PIGO_ONE_LOCAL1=0" "local2:PIGO_ONE_LOCAL0=2 PIGO_ONE_LOCAL1=2"
   "pti5:PIGO_ONE_PTI=5" "cs3:PIGO_ONE_BIG_CS=3" "nhb4:PIGO_ONE_NH_BIG=4" "w20:PIGO_ONE_W1_X10=20" "w45:PIGO_ONE_W1_X10=45" "s224:PIGO_ONE_SLOTS=224" "s320:PIGO_ONE_SLOTS=320"
   "c256:PIGO_REG_CHUNK0=256" "c128:PIGO_REG_CHUNK0=128 PIGO_REG_CHUNK1=64" "one_b:")
timeout 400 python scripts/ab.py --frames 1 --steps 100 --no-cluster --kernel-times "${S[@]}" 2>$O/ab_one.err | tee $O/ab_one.txt || tail -5 $O/ab_one.err
timeout 200 python scripts/ab.py --frames 1 --steps 100 --no-cluster --kind noise "one:" "v2:PIGO_SCAN_VARIANT=2" 2>$O/ab_one_noise.err | tee $O/ab_one_noise.txt || tail -5 $O/ab_one_noise.err
timeout 200 python scripts/ab.py --frames 1 --steps 100 --no-cluster --angle 0.8 "one:" "v2:PIGO_SCAN_VARIANT=2" 2>$O/ab_one_rot.err | tee $O/ab_one_rot.txt || tail -5 $O/ab_one_rot.err
timeout 200 python scripts/ab.py --frames 3 --steps 50 --no-cluster "one:" "v2:PIGO_SCAN_VARIANT=2" 2>$O/ab_three.err | tee $O/ab_three.txt || tail -5 $O/ab_three.err
export PIGO_HIP_LIB=$GRAFT_REPO_ROOT/pigo_amd/csrc/libpigo_hip_debug.so
timeout 120 env PIGO_SYNC_DEBUG=1 python scripts/one_trace.py 2>$O/trace.err | tee $O/trace.txt || tail -5 $O/trace.err
grep "k_scan_one" $O/trace.err | head -2
timeout 120 python scripts/one_trace.py --kind noise 2>>$O/trace.err | tee -a $O/trace.txt
timeout 120 python scripts/one_trace.py --rows 400 --cols 320 --shift 0.2 2>>$O/trace.err | tee -a $O/trace.txt
unset PIGO_HIP_LIB PIGO_TUNING
timeout 200 python scripts/single_frame_latency.py 2>&1 | tail -2 | tee $O/single.txt
