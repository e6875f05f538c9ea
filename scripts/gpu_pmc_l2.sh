cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_l2
B="python bench.py --frames 64 --steps 3 --warmup 1 --no-cpu-baseline --no-cluster"
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum -d gpurun_out/prof_l2/a -o p -- $B > gpurun_out/prof_l2/a.log 2>&1; echo "a rc=$?"
timeout 600 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d gpurun_out/prof_l2/b -o p -- $B > gpurun_out/prof_l2/b.log 2>&1; echo "b rc=$?"; tail -3 gpurun_out/prof_l2/b.log
