# fabric-side traffic of the 4K stress config (BASELINE configs[4]) on a 64-frame step (531 MB of frames: twice the Infinity Cache)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r6_4k; mkdir -p $O
COMMIT="${COMMIT:-unknown}"
B="python bench.py --rows 2160 --cols 3840 --min-size 20 --max-size 2000 --shift 0.05 --scale 1.05 --frames 64 --det-cap 32768 --gather-cap 64 --steps 2 --warmup 1 --no-cpu-baseline --no-gray --no-single-frame --no-config-legs --shard-frames 0 --verify-frames 0 --no-kernel-times"
timeout 900 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o p -- $B > $O/pmc_fetch.log 2>&1; echo "pmc_fetch rc=$?"
timeout 900 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o p -- $B > $O/pmc_write.log 2>&1; echo "pmc_write rc=$?"
timeout 900 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum -d $O/pmc_tcc -o p -- $B > $O/pmc_tcc.log 2>&1; echo "pmc_tcc rc=$?"
timeout 900 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum -d $O/pmc_ea -o p -- $B > $O/pmc_ea.log 2>&1; echo "pmc_ea rc=$?"
python scripts/make_traffic.py $(find $O/pmc_fetch -name "*.db" | head -1) $(find $O/pmc_write -name "*.db" | head -1) 3 64 $(find $O/pmc_tcc -name "*.db" | head -1) "$COMMIT" $(find $O/pmc_ea -name "*.db" | head -1) > $O/traffic_4k.json 2>$O/traffic_4k.err; echo "traffic rc=$?"
grep -E "fabric_bytes_per_frame\"|hit_rate|ea_read_bytes" $O/traffic_4k.json
timeout 600 $B 2>/dev/null | cut -c1-200
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_tcc $O/pmc_ea
