# Round 5, call 15: reg_walk with two half-batches half a level apart (PIGO_OPT_SKEW) against the lock-step walk
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5p; mkdir -p $O
export PIGO_TUNING=1
for rep in 1 2; do
for lib in libpigo_hip.so libpigo_hip_noskew.so; do
  export PIGO_HIP_LIB=$GRAFT_REPO_ROOT/pigo_amd/csrc/$lib
  timeout 300 python scripts/ab.py --frames 128 --steps 10 --kernel-times "$lib:" 2>$O/ab.err | tail -1 | tee -a $O/ab.txt || tail -5 $O/ab.err
done
done
for lib in libpigo_hip.so libpigo_hip_noskew.so; do
  export PIGO_HIP_LIB=$GRAFT_REPO_ROOT/pigo_amd/csrc/$lib
  timeout 300 python scripts/ab.py --frames 1 --steps 100 --no-cluster "$lib:" 2>$O/ab.err | tail -1 | tee -a $O/ab.txt || tail -5 $O/ab.err
  timeout 300 python scripts/ab.py --frames 64 --steps 10 --angle 0.8 "$lib:" 2>$O/ab.err | tail -1 | tee -a $O/ab.txt || tail -5 $O/ab.err
done
