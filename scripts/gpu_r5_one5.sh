# Round 5, call 5: the hand-off timeout of call 4 -- plain-store zeroing of the counters (as in call 4) against agent-scope stores, five processes each
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5e; mkdir -p $O
export PIGO_TUNING=1
for i in 1 2 3 4 5; do
  timeout 100 python scripts/ab.py --frames 1 --steps 100 --no-cluster "plain$i:PIGO_ONE_PLAINZERO=1" 2>$O/e.err | tail -1 || tail -1 $O/e.err
done 2>&1 | tee $O/plain.txt
for i in 1 2 3 4 5; do
  timeout 100 python scripts/ab.py --frames 1 --steps 100 --no-cluster "sc1_$i:" 2>$O/e.err | tail -1 || tail -1 $O/e.err
done 2>&1 | tee $O/sc1.txt
S=("one:" "v2:PIGO_SCAN_VARIANT=2" "norestore:PIGO_ONE_RESTORE=0" "late0:PIGO_ONE_LATE_ITEMS=0" "late32:PIGO_ONE_LATE_ITEMS=32" "late96:PIGO_ONE_LATE_ITEMS=96" "ntl2:PIGO_ONE_NT_LATE=2"
   "d64:PIGO_ONE_DEEP0=64 PIGO_ONE_DEEP1=64" "d128:PIGO_ONE_DEEP0=128 PIGO_ONE_DEEP1=128" "d256:PIGO_ONE_DEEP0=256 PIGO_ONE_DEEP1=256"
   "noquad:PIGO_REG_QUAD0=0 PIGO_REG_QUAD1=0" "mid28:PIGO_NH_REG1=28" "w20:PIGO_ONE_W1_X10=20" "w45:PIGO_ONE_W1_X10=45" "s240:PIGO_ONE_SLOTS=240" "local2:PIGO_ONE_LOCAL0=2 PIGO_ONE_LOCAL1=2" "one_b:")
timeout 400 python scripts/ab.py --frames 1 --steps 100 --no-cluster --kernel-times "${S[@]}" 2>$O/ab_one.err | tee $O/ab_one.txt || tail -5 $O/ab_one.err
timeout 200 python scripts/ab.py --frames 1 --steps 100 --no-cluster --kind noise "one:" "v2:PIGO_SCAN_VARIANT=2" "d64:PIGO_ONE_DEEP0=64 PIGO_ONE_DEEP1=64" 2>$O/ab_one_noise.err | tee $O/ab_one_noise.txt || tail -5 $O/ab_one_noise.err
