# round-1 baseline artefacts: default bench line (with cpu_baseline), rocprofv3 kernel trace, PMC passes
mkdir -p gpurun_out/prof_r1 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"; cat gpurun_out/bench_default.json
B="python bench.py --frames 32 --steps 3 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r1/trace -o r1 -- $B > gpurun_out/prof_r1/trace.log 2>&1; echo "trace rc=$?"
rocprofv3 -L > gpurun_out/prof_r1/counters_list.txt 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum -d gpurun_out/prof_r1/pmc1 -o p1 -- $B > gpurun_out/prof_r1/pmc1.log 2>&1; echo "pmc1 rc=$?"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d gpurun_out/prof_r1/pmc2 -o p2 -- $B > gpurun_out/prof_r1/pmc2.log 2>&1; echo "pmc2 rc=$?"
timeout 600 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr GRBM_GUI_ACTIVE -d gpurun_out/prof_r1/pmc3 -o p3 -- $B > gpurun_out/prof_r1/pmc3.log 2>&1; echo "pmc3 rc=$?"
find gpurun_out/prof_r1 -name "*.csv" | head -30; du -sh gpurun_out/prof_r1
