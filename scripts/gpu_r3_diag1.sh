# Round-3 diagnostics: (1) TA microbenchmark, (2) counter list, (3) L2 hit/miss + TA/TCP counters of today's big-scale kernels,
# (4) the round-2 abort reproduced without the build mutex under rocgdb
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
[ -x scripts/micro/ta_gather ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/ta_gather.hip -o scripts/micro/ta_gather
timeout 300 scripts/micro/ta_gather 2000 > $O/ta_gather.txt 2>&1; echo "ta_gather rc=$?"; head -3 $O/ta_gather.txt
(rocprofv3 -L 2>&1 || rocprofv3-avail list 2>&1) > $O/counters_all.txt; grep -o -E "\b(TCC|TCP|TA|TD|SQ|GRBM)_[A-Za-z0-9_]+" $O/counters_all.txt | sort -u > $O/counters.txt; wc -l $O/counters.txt
B="env PIGO_SIDE_STREAM=0 python bench.py --frames 64 --steps 3 --warmup 1 --no-cpu-baseline --no-single-frame --no-gray --shard-frames 0 --verify-frames 0"
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum" "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum TCP_GATE_EN1_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set -d $O/pmc_$n -o p -- $B > $O/pmc_$n.log 2>&1; echo "pmc $n rc=$?"
done
python scripts/summarize_prof.py "r3 diag1: $B" "" $O/pmc_*/p_results.db > $O/diag1_pmc.txt 2>$O/diag1_pmc.err; grep -E "k_scan_tile|k_tail_deep<false, false, false>|k_scan_region" $O/diag1_pmc.txt | cut -c1-140
rm -rf $O/pmc_*/
# (4) the abort: four threads building slots concurrently, no build mutex
for i in 1 2 3; do
  PIGO_NO_BUILD_MU=1 timeout 300 rocgdb -batch -ex "handle SIGABRT stop print" -ex "handle SIGSEGV stop print" -ex run -ex "thread apply all bt 25" --args python -m pytest tests/test_gpu_parity.py -q -x -k reentrant -p no:cacheprovider > $O/abort_gdb_$i.txt 2>&1
  echo "gdb run $i rc=$? $(grep -c -E 'SIGABRT|SIGSEGV' $O/abort_gdb_$i.txt) signals; $(grep -E 'passed|failed' $O/abort_gdb_$i.txt | tail -1)"
done
grep -h -B2 -A30 -E "received signal" $O/abort_gdb_*.txt | head -150 > $O/abort_bt.txt; head -60 $O/abort_bt.txt | cut -c1-220
