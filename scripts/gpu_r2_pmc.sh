# PMC passes over a short bench run:  bash scripts/gpu_r2_pmc.sh NAME "ENV=.. ENV2=.." "bench args"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
name="$1"; envs="$2"; args="$3"
T=gpurun_out/r2/pmc_$name; mkdir -p $T
B="env PIGO_PIPE_CHUNKS=1 PIGO_SIDE_STREAM=0 $envs python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gray --shard-frames 0 --verify-frames 0 $args"
timeout 600 rocprofv3 --kernel-trace --stats -d $T/trace -o t -- $B > $T/trace.log 2>&1; echo "trace rc=$?"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $T/pmc_sq -o p -- $B > $T/pmc_sq.log 2>&1; echo "pmc_sq rc=$?"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $T/pmc_sq2 -o p -- $B > $T/pmc_sq2.log 2>&1; echo "pmc_sq2 rc=$?"
python scripts/summarize_prof.py "$name: $B" $T/trace/t_results.db $T/pmc_sq/p_results.db $T/pmc_sq2/p_results.db > gpurun_out/r2/pmc_$name.txt 2> gpurun_out/r2/pmc_$name.err; echo "summary rc=$?"
grep -v "at::\|rocclr\|k_build\|k_restore\|k_sort\|k_cluster\|k_gosort" gpurun_out/r2/pmc_$name.txt | cut -c1-140
rm -rf $T
