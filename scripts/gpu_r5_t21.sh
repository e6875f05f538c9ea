# Round 5, call 21: persistent k_scan_region with the next region's first units prefetched into registers (PIGO_REG_PERSIST)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5v; mkdir -p $O
export PIGO_TUNING=1
timeout 400 python scripts/ab.py --frames 128 --steps 10 --kernel-times "persist:" "oneshot:PIGO_REG_PERSIST=0" "persist_b:" "oneshot_b:PIGO_REG_PERSIST=0" 2>$O/ab.err | tee $O/ab.txt || tail -5 $O/ab.err
timeout 300 python scripts/ab.py --frames 64 --steps 10 --angle 0.8 "persist:" "oneshot:PIGO_REG_PERSIST=0" 2>$O/ab.err | tee -a $O/ab.txt || tail -5 $O/ab.err
