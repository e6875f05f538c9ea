# Round 5, call 14: LDS microbenchmark (slope timing), counter list, the suite without PIGO_TUNING, a few batch A/Bs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5o; mkdir -p $O
timeout 120 scripts/micro/lds_gather 400 | tee $O/lds_gather.txt
rocprofv3 -L 2>/dev/null | grep -i -E "dram|mall|hbm|EA0_RDREQ|EA0_WRREQ|EA_RDREQ|EA_WRREQ|MC_RD|MC_WR" | head -60 > $O/counters.txt; wc -l $O/counters.txt; head -40 $O/counters.txt
timeout 900 python -m pytest tests/test_gpu_production_env.py -m gpu -x -q 2>&1 | tail -30 | tee $O/pytest_prod.txt
export PIGO_TUNING=1
S=("base:" "deep1_64:PIGO_REG_DEEP1=64" "deep1_128:PIGO_REG_DEEP1=128" "deep1_256:PIGO_REG_DEEP1=256" "prio2:PIGO_REG_PRIO=2" "base_b:")
timeout 600 python scripts/ab.py --frames 128 --steps 10 "${S[@]}" 2>$O/ab.err | tee $O/ab.txt || tail -5 $O/ab.err
