# Round 5, call 8: per-item phase breakdown of one k_scan_one launch (debug library)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5h; mkdir -p $O
export PIGO_HIP_LIB=$GRAFT_REPO_ROOT/pigo_amd/csrc/libpigo_hip_debug.so
timeout 120 python scripts/one_trace.py 2>$O/trace.err | tee $O/trace.txt || tail -5 $O/trace.err
timeout 120 python scripts/one_trace.py --kind noise 2>>$O/trace.err | tee -a $O/trace.txt
timeout 120 python scripts/one_trace.py PIGO_ONE_SLOTS=384 2>>$O/trace.err | tee -a $O/trace.txt
