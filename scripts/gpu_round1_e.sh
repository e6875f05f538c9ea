mkdir -p gpurun_out/prof_r1e && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --frames 32 --steps 2 --warmup 1 --no-cpu-baseline --no-cluster"
export PIGO_TILE_RULES="6,32,16384;6,16,24576;6,8,36864"
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r1e/trace -o t -- $B > gpurun_out/prof_r1e/trace.log 2>&1; echo "trace rc=$?"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d gpurun_out/prof_r1e/pmc1 -o p -- $B > gpurun_out/prof_r1e/pmc1.log 2>&1; echo "pmc1 rc=$?"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE -d gpurun_out/prof_r1e/pmc2 -o p -- $B > gpurun_out/prof_r1e/pmc2.log 2>&1; echo "pmc2 rc=$?"
ls -R gpurun_out/prof_r1e | head -20
