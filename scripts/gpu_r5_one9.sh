# Round 5, call 9: regions that end late hand their whole deep list to the idle consumers (PIGO_ONE_OFFLOAD_X10); survivors of the local passes pushed together
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5i; mkdir -p $O
export PIGO_TUNING=1
S=("one:" "off0:PIGO_ONE_OFFLOAD_X10=0" "off5:PIGO_ONE_OFFLOAD_X10=5" "off10:PIGO_ONE_OFFLOAD_X10=10" "off40:PIGO_ONE_OFFLOAD_X10=40" "off80:PIGO_ONE_OFFLOAD_X10=80" "off1:PIGO_ONE_OFFLOAD_X10=1"
   "off10_nt2:PIGO_ONE_OFFLOAD_X10=10 PIGO_ONE_NT=2" "off10_ntl2:PIGO_ONE_OFFLOAD_X10=10 PIGO_ONE_NT_LATE=2" "off10_late64:PIGO_ONE_OFFLOAD_X10=10 PIGO_ONE_LATE_ITEMS=64" "off10_late200:PIGO_ONE_OFFLOAD_X10=10 PIGO_ONE_LATE_ITEMS=200" "one_b:")
timeout 500 python scripts/ab.py --frames 1 --steps 100 --no-cluster --kernel-times "${S[@]}" 2>$O/ab_one.err | tee $O/ab_one.txt || tail -5 $O/ab_one.err
timeout 200 python scripts/ab.py --frames 1 --steps 100 --no-cluster --kind noise "one:" "off0:PIGO_ONE_OFFLOAD_X10=0" "off5:PIGO_ONE_OFFLOAD_X10=5" 2>$O/ab_one_noise.err | tee $O/ab_one_noise.txt || tail -5 $O/ab_one_noise.err
timeout 200 python scripts/ab.py --frames 1 --steps 100 --no-cluster --angle 0.8 "one:" "off0:PIGO_ONE_OFFLOAD_X10=0" 2>$O/ab_one_rot.err | tee $O/ab_one_rot.txt || tail -5 $O/ab_one_rot.err
timeout 200 python scripts/ab.py --frames 3 --steps 50 --no-cluster "one:" "off0:PIGO_ONE_OFFLOAD_X10=0" 2>$O/ab_three.err | tee $O/ab_three.txt || tail -5 $O/ab_three.err
export PIGO_HIP_LIB=$GRAFT_REPO_ROOT/pigo_amd/csrc/libpigo_hip_debug.so
timeout 120 python scripts/one_trace.py 2>$O/trace.err | tee $O/trace.txt || tail -5 $O/trace.err
unset PIGO_HIP_LIB
timeout 300 python scripts/one_stress.py --launches 2000 --sizes 1080x1920,720x1280 2>$O/stress.err | tee $O/stress.txt || tail -5 $O/stress.err
