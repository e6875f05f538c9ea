# the rocprofv3 passes of gpu_round1_final.sh only
# Round-1 measurement: parity suite, smoke, the default bench line, the RCCL path at world size 1, rocprofv3 trace + PMC passes
mkdir -p gpurun_out/prof_r1f && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="env PIGO_PIPE_CHUNKS=1 PIGO_SIDE_STREAM=0 python bench.py --frames 64 --steps 5 --warmup 2 --no-cpu-baseline --no-single-frame"
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r1f/trace -o t -- $B > gpurun_out/prof_r1f/trace.log 2>&1; echo "trace rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof_r1f/pmc_fetch -o p -- $B > gpurun_out/prof_r1f/pmc_fetch.log 2>&1; echo "pmc_fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof_r1f/pmc_write -o p -- $B > gpurun_out/prof_r1f/pmc_write.log 2>&1; echo "pmc_write rc=$?"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d gpurun_out/prof_r1f/pmc_sq -o p -- $B > gpurun_out/prof_r1f/pmc_sq.log 2>&1; echo "pmc_sq rc=$?"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/prof_r1f/pmc_sq2 -o p -- $B > gpurun_out/prof_r1f/pmc_sq2.log 2>&1; echo "pmc_sq2 rc=$?"
T=gpurun_out/prof_r1f
python scripts/summarize_prof.py "round 1 final (see scripts/gpu_round1_final.sh): python bench.py --frames 64 --steps 5 --warmup 2 --no-cpu-baseline -- 64 x 1080p SYN-FACES frames per step; 12 scan steps per run (2 warm-up + 5 timed + 5 per-kernel event reps), PIGO_PIPE_CHUNKS=1 PIGO_SIDE_STREAM=0 (no chunking, no side stream) so that every launch is one un-overlapped 64-frame batch like bench.py kernel_ms + the gray / puploc side legs" $T/trace/t_results.db $T/pmc_fetch/p_results.db $T/pmc_write/p_results.db $T/pmc_sq/p_results.db $T/pmc_sq2/p_results.db > gpurun_out/final_summary.txt 2>gpurun_out/final_summary.err; echo "summary rc=$?"; head -22 gpurun_out/final_summary.txt | cut -c1-150
python scripts/make_traffic.py $T/pmc_fetch/p_results.db $T/pmc_write/p_results.db 12 64 > gpurun_out/traffic.json 2>gpurun_out/traffic.err; echo "traffic rc=$?"; grep hbm_bytes gpurun_out/traffic.json
rm -rf $T/trace $T/pmc_fetch $T/pmc_write $T/pmc_sq $T/pmc_sq2
