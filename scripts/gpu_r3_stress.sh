cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
for m in seq two_build build_only nocapture full full_own_stream; do
  timeout 120 python scripts/gpu_r3_stress.py $m > $O/stress_$m.log 2>&1; echo "$m rc=$? $(grep -E '^(seq|two_build|build_only|nocapture|full)' $O/stress_$m.log | cut -c1-300) $(grep -m1 -E 'Memory access fault|Aborted|HSA_STATUS' $O/stress_$m.log | cut -c1-200)"
done
