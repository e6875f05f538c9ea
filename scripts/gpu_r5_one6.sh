# Round 5, call 6: the claim-before-done ordering fix of k_scan_one's consumers -- GPU suite, hand-off stress (alone and next to a live batch plan), the A/B list that failed in call 5
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5f; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest.txt
timeout 500 python scripts/one_stress.py --launches 3000 2>$O/stress.err | tee $O/stress.txt || tail -5 $O/stress.err
timeout 200 python scripts/one_stress.py --launches 1500 --sizes 1080x1920 PIGO_ONE_LATE_ITEMS=32 2>>$O/stress.err | tee -a $O/stress.txt
timeout 200 python scripts/one_stress.py --launches 1500 --sizes 1080x1920 --frames 3 2>>$O/stress.err | tee -a $O/stress.txt
timeout 200 python scripts/one_stress.py --launches 1500 --sizes 1080x1920 --angle 0.8 2>>$O/stress.err | tee -a $O/stress.txt
export PIGO_TUNING=1
S=("one:" "v2:PIGO_SCAN_VARIANT=2" "late32:PIGO_ONE_LATE_ITEMS=32" "late96:PIGO_ONE_LATE_ITEMS=96" "ntl2:PIGO_ONE_NT_LATE=2"
   "d64:PIGO_ONE_DEEP0=64 PIGO_ONE_DEEP1=64" "d128:PIGO_ONE_DEEP0=128 PIGO_ONE_DEEP1=128" "d256:PIGO_ONE_DEEP0=256 PIGO_ONE_DEEP1=256"
   "noquad:PIGO_REG_QUAD0=0 PIGO_REG_QUAD1=0" "mid28:PIGO_NH_REG1=28" "w20:PIGO_ONE_W1_X10=20" "w45:PIGO_ONE_W1_X10=45" "s240:PIGO_ONE_SLOTS=240" "local2:PIGO_ONE_LOCAL0=2 PIGO_ONE_LOCAL1=2"
   "local0:PIGO_ONE_LOCAL0=0 PIGO_ONE_LOCAL1=0" "pti5:PIGO_ONE_PTI=5" "nt2:PIGO_ONE_NT=2" "one_b:")
timeout 500 python scripts/ab.py --frames 1 --steps 100 --no-cluster --kernel-times "${S[@]}" 2>$O/ab_one.err | tee $O/ab_one.txt || tail -5 $O/ab_one.err
timeout 200 python scripts/ab.py --frames 1 --steps 100 --no-cluster --kind noise "one:" "v2:PIGO_SCAN_VARIANT=2" "d64:PIGO_ONE_DEEP0=64 PIGO_ONE_DEEP1=64" 2>$O/ab_one_noise.err | tee $O/ab_one_noise.txt || tail -5 $O/ab_one_noise.err
