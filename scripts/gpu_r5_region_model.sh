# Round 5: a measured bound for k_scan_region -- (a) the LDS microbenchmark of its byte gathers, (b) per-stage cost and SQ counters by elimination
# (debug library, PIGO_REG_CUT), each scale group's launch alone on the chip
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5m; mkdir -p $O
[ -x scripts/micro/lds_gather ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/lds_gather.hip -o scripts/micro/lds_gather
timeout 120 scripts/micro/lds_gather 400 | tee $O/lds_gather.txt
export PIGO_TUNING=1 PIGO_HIP_LIB=$GRAFT_REPO_ROOT/pigo_amd/csrc/libpigo_hip_debug.so PIGO_BIG_SKIP=3
B="python bench.py --frames 64 --steps 5 --warmup 2 --no-cpu-baseline --no-gray --shard-frames 0 --no-single-frame --no-config-legs --verify-frames 0 --no-kernel-times --no-cluster"
for g in 0 1; do
  for c in 1 2 3 4 0; do
    export PIGO_REG_ONLY=$g PIGO_REG_CUT=$c
    timeout 300 rocprofv3 --kernel-trace --stats -d $O/g${g}_cut${c}_trace -o t -- $B > $O/g${g}_cut${c}_trace.log 2>&1; echo "g$g cut$c trace rc=$?"
    timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $O/g${g}_cut${c}_pmc -o p -- $B > $O/g${g}_cut${c}_pmc.log 2>&1; echo "g$g cut$c pmc rc=$?"
  done
done
python scripts/region_stages.py $O 64 | tee $O/region_stages.txt
rm -rf $O/g*_trace $O/g*_pmc
