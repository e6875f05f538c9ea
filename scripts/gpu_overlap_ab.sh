# timeline of one timed step (side chain next to the region launches) for each library given: bash scripts/gpu_overlap_ab.sh OUTDIR lib1.so lib2.so ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$1; shift; mkdir -p $O
T="python bench.py --frames 128 --steps 6 --warmup 2 --no-cpu-baseline --no-gray --no-single-frame --no-config-legs --shard-frames 0 --verify-frames 0 --no-kernel-times"
for L in "$@"; do
  N=$(basename $L .so)
  env PIGO_HIP_LIB=$GRAFT_REPO_ROOT/pigo_amd/csrc/$L timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_$N -o t -- $T > $O/trace_$N.log 2>&1; echo "trace $N rc=$?"
  python scripts/trace_overlap.py $(find $O/trace_$N -name "*.db" | head -1) 2 > $O/overlap_$N.txt 2>&1; tail -14 $O/overlap_$N.txt
  rm -rf $O/trace_$N
done
