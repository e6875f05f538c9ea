# Round-6 measurement: parity suite (incl. the run without PIGO_TUNING), smoke, the one-frame launch (trace + timeline), the OVERLAPPED step's timeline,
# per-kernel trace + PMC of the serialised step, fabric / DRAM-side traffic on a 512-frame step, the bench lines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r6f; mkdir -p $O
COMMIT="${COMMIT:-unknown}"
nproc > $O/host.txt; (rocminfo | grep -m3 "Marketing Name" ) >> $O/host.txt 2>&1
if [ -z "$SKIP_PYTEST" ]; then timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -1 | tee $O/pytest_gpu_tail.txt; fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
# (0) one frame per call: kernel trace of 100 + 20 calls, timeline of one launch (debug library)
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_one -o t -- python scripts/single_frame_latency.py > $O/trace_one.log 2>&1; echo "trace_one rc=$?"
python scripts/summarize_prof.py "round 6 (scripts/gpu_round6_final.sh, commit $COMMIT): python scripts/single_frame_latency.py -- ONE 1080p frame per call: 23 x pigo_run_cascade (host buffer) + 105 x pigo_plan_run (HBM-resident); every call is ONE k_scan_one launch" $(find $O/trace_one -name "*.db" | head -1) > $O/single_frame_trace.txt 2>$O/single_frame_trace.err; head -8 $O/single_frame_trace.txt | cut -c1-120
grep "single 1080p" $O/trace_one.log | tee $O/single_frame_profiled.txt
python scripts/single_frame_latency.py 2>&1 | grep "single 1080p" | tee $O/single_frame.txt
[ -f pigo_amd/csrc/libpigo_hip_debug.so ] || python -m pigo_amd.build --debug > /dev/null 2>&1
(env PIGO_HIP_LIB=$GRAFT_REPO_ROOT/pigo_amd/csrc/libpigo_hip_debug.so timeout 120 python scripts/one_trace.py; env PIGO_HIP_LIB=$GRAFT_REPO_ROOT/pigo_amd/csrc/libpigo_hip_debug.so timeout 120 python scripts/one_trace.py --kind noise) > $O/one_frame_timeline.txt 2>$O/one_frame_timeline.err; head -12 $O/one_frame_timeline.txt | cut -c1-160
timeout 400 python scripts/one_stress.py --launches 2000 --sizes 1080x1920,720x1280 > $O/one_stress.txt 2>&1; tail -4 $O/one_stress.txt
# (1) the step as it is timed: side chain next to the region launches -- begin / end timestamps of one step
T="python bench.py --frames 128 --steps 6 --warmup 2 --no-cpu-baseline --no-gray --no-single-frame --no-config-legs --shard-frames 0 --verify-frames 0 --no-kernel-times"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_ov -o t -- $T > $O/trace_ov.log 2>&1; echo "trace_ov rc=$?"
python scripts/trace_overlap.py $(find $O/trace_ov -name "*.db" | head -1) 2 > $O/overlap_timeline.txt 2>&1; tail -12 $O/overlap_timeline.txt
# (2) every launch alone on one stream (what bench.py's kernel_ms reports), 64 frames per step, + SQ PMC passes of the same command
B="env PIGO_TUNING=1 PIGO_SIDE_STREAM=0 python bench.py --frames 64 --steps 5 --warmup 2 --no-cpu-baseline --no-gray --no-single-frame --no-config-legs --shard-frames 0 --verify-frames 0"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- $B > $O/trace.log 2>&1; echo "trace rc=$?"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc_sq -o p -- $B > $O/pmc_sq.log 2>&1; echo "pmc_sq rc=$?"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O/pmc_sq2 -o p -- $B > $O/pmc_sq2.log 2>&1; echo "pmc_sq2 rc=$?"
# (3) traffic on a 512-frame step (1.06 GB of frames resident: four times the Infinity Cache), the step as it is timed (side chain next to the regions)
B5="python bench.py --frames 512 --steps 3 --warmup 1 --no-cpu-baseline --no-gray --no-single-frame --no-config-legs --shard-frames 0 --verify-frames 0 --no-kernel-times"
timeout 900 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o p -- $B5 > $O/pmc_fetch.log 2>&1; echo "pmc_fetch rc=$?"
timeout 900 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o p -- $B5 > $O/pmc_write.log 2>&1; echo "pmc_write rc=$?"
timeout 900 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum -d $O/pmc_tcc -o p -- $B5 > $O/pmc_tcc.log 2>&1; echo "pmc_tcc rc=$?"
timeout 900 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum -d $O/pmc_ea -o p -- $B5 > $O/pmc_ea.log 2>&1; echo "pmc_ea rc=$?"
python scripts/summarize_prof.py "round 6 final (scripts/gpu_round6_final.sh, commit $COMMIT): $B -- 64 x 1080p SYN-FACES frames per step; 12 scan steps per run (2 warm-up + 5 timed + 5 per-kernel event reps); PIGO_SIDE_STREAM=0 puts the side chain (k_scan_big -> k_big_pool -> k_tail_deep of the big scales) behind the region launches on ONE stream so that every launch is un-overlapped like bench.py's kernel_ms (the timed default runs it NEXT to the first region launch: profiles/r05_overlap_timeline.txt).  Traffic passes: $B5 (4 steps per run, the step as it is timed)" $(find $O/trace -name "*.db" | head -1) $(find $O/pmc_sq -name "*.db" | head -1) $(find $O/pmc_sq2 -name "*.db" | head -1) $(find $O/pmc_fetch -name "*.db" | head -1) $(find $O/pmc_write -name "*.db" | head -1) $(find $O/pmc_tcc -name "*.db" | head -1) $(find $O/pmc_ea -name "*.db" | head -1) > $O/final_summary.txt 2>$O/final_summary.err; echo "summary rc=$?"; head -14 $O/final_summary.txt | cut -c1-150
python scripts/make_traffic.py $(find $O/pmc_fetch -name "*.db" | head -1) $(find $O/pmc_write -name "*.db" | head -1) 4 512 $(find $O/pmc_tcc -name "*.db" | head -1) "$COMMIT" $(find $O/pmc_ea -name "*.db" | head -1) > $O/traffic.json 2>$O/traffic.err; echo "traffic rc=$?"; grep -E "fabric_bytes_per_frame\"|hit_rate|dram_destined|ea_read_bytes" $O/traffic.json
# the traffic figure bench.py cites is the one just measured (profiles/r06_traffic.json carries its commit)
if [ -s $O/traffic.json ] && grep -q fabric_bytes_per_frame $O/traffic.json; then cp $O/traffic.json profiles/r06_traffic.json; fi
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-400 $O/bench_default.json; tail -2 $O/bench_default.err
C="--no-cpu-baseline --no-gray --shard-frames 0 --no-config-legs"
timeout 300 python bench.py --kind noise $C > $O/bench_noise.json 2> $O/bench_noise.err; echo "noise rc=$?"; cut -c1-160 $O/bench_noise.json
timeout 300 python bench.py --angle 0.8 $C > $O/bench_rot.json 2> $O/bench_rot.err; echo "rot rc=$?"; cut -c1-160 $O/bench_rot.json
timeout 300 python bench.py --rows 2160 --cols 3840 --min-size 20 --max-size 2000 --shift 0.05 --scale 1.05 --frames 8 --det-cap 32768 --gather-cap 64 --steps 5 --warmup 2 $C --verify-frames 2 > $O/bench_4k.json 2> $O/bench_4k.err; echo "4k rc=$?"; cut -c1-160 $O/bench_4k.json; tail -2 $O/bench_4k.err
timeout 300 python bench.py --frames 1 --steps 20 --warmup 5 $C --verify-frames 1 --no-single-frame > $O/bench_1frame.json 2> $O/bench_1frame.err; echo "1frame rc=$?"; cut -c1-300 $O/bench_1frame.json
# the C ABI's collective at world size 1 through a real RCCL communicator
timeout 300 python bench.py --force-dist --frames 32 --steps 3 --warmup 1 $C > $O/bench_dist1.json 2> $O/bench_dist1.err; echo "dist1 rc=$?"; cut -c1-160 $O/bench_dist1.json
# region phase timers (debug build): the small group alone and with the side chain next to it
(echo "## small group"; REG_ONLY=0 bash scripts/gpu_r4_phases.sh; echo "## mid group"; REG_ONLY=1 PHASE_SPECS="alone:PIGO_BIG_SKIP=3 with_side:PIGO_X=1" bash scripts/gpu_r4_phases.sh) > $O/region_phases.txt 2>&1; tail -8 $O/region_phases.txt
# round 6: the two microbenchmarks behind DESIGN.md section 4, and the 4K step's timeline
timeout 300 scripts/micro/valu_rate 2000 > $O/valu_rate.txt 2>&1; echo "valu_rate rc=$?"
timeout 600 scripts/micro/lds_valu_mix 300 > $O/lds_valu_mix.txt 2>&1; echo "lds_valu_mix rc=$?"
bash scripts/gpu_overlap_4k.sh $O > $O/overlap_4k.log 2>&1; tail -3 $O/overlap_4k.txt
rm -rf $O/trace $O/trace_ov $O/trace_one $O/pmc_fetch $O/pmc_write $O/pmc_tcc $O/pmc_ea $O/pmc_sq $O/pmc_sq2
du -sh $O
