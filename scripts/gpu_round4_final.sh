# Round-4 measurement (traffic first: the bench line cites it): parity suite, smoke, the default bench line (with its config legs), the other configs on their own,
# rocprofv3 kernel trace of the OVERLAPPED step (begin/end timestamps) and of the serialised one, PMC passes, traffic, phase timers
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_r4f
O=gpurun_out/prof_r4f
COMMIT="${COMMIT:-unknown}"
nproc > $O/host.txt; (rocminfo | grep -m3 "Marketing Name" ) >> $O/host.txt 2>&1
if [ -z "$SKIP_PYTEST" ]; then timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -1 | tee $O/pytest_gpu_tail.txt; fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
# (1) the step as it is timed: side chain next to the region launches -- begin / end timestamps of one step
T="python bench.py --frames 128 --steps 6 --warmup 2 --no-cpu-baseline --no-gray --no-single-frame --no-config-legs --shard-frames 0 --verify-frames 0 --no-kernel-times"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_ov -o t -- $T > $O/trace_ov.log 2>&1; echo "trace_ov rc=$?"
python scripts/trace_overlap.py $(find $O/trace_ov -name "*.db" | head -1) 2 > $O/overlap_timeline.txt 2>&1; tail -12 $O/overlap_timeline.txt
# (2) every launch alone on one stream (what bench.py's kernel_ms reports), 64 frames per step, + PMC passes of the same command
B="env PIGO_TUNING=1 PIGO_SIDE_STREAM=0 python bench.py --frames 64 --steps 5 --warmup 2 --no-cpu-baseline --no-gray --no-single-frame --no-config-legs --shard-frames 0 --verify-frames 0"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- $B > $O/trace.log 2>&1; echo "trace rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o p -- $B > $O/pmc_fetch.log 2>&1; echo "pmc_fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o p -- $B > $O/pmc_write.log 2>&1; echo "pmc_write rc=$?"
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum -d $O/pmc_tcc -o p -- $B > $O/pmc_tcc.log 2>&1; echo "pmc_tcc rc=$?"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc_sq -o p -- $B > $O/pmc_sq.log 2>&1; echo "pmc_sq rc=$?"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O/pmc_sq2 -o p -- $B > $O/pmc_sq2.log 2>&1; echo "pmc_sq2 rc=$?"
python scripts/summarize_prof.py "round 4 final (scripts/gpu_round4_final.sh, commit $COMMIT): $B -- 64 x 1080p SYN-FACES frames per step; 12 scan steps per run (2 warm-up + 5 timed + 5 per-kernel event reps); PIGO_SIDE_STREAM=0 puts the side chain (k_scan_big + k_tail_deep of the big scales) behind the region launches on ONE stream so that every launch is un-overlapped like bench.py's kernel_ms (the timed default runs it NEXT to the first region launch: profiles/r04_overlap_timeline.txt)" $(find $O/trace -name "*.db" | head -1) $(find $O/pmc_fetch -name "*.db" | head -1) $(find $O/pmc_write -name "*.db" | head -1) $(find $O/pmc_tcc -name "*.db" | head -1) $(find $O/pmc_sq -name "*.db" | head -1) $(find $O/pmc_sq2 -name "*.db" | head -1) > $O/final_summary.txt 2>$O/final_summary.err; echo "summary rc=$?"; head -14 $O/final_summary.txt | cut -c1-150
python scripts/make_traffic.py $(find $O/pmc_fetch -name "*.db" | head -1) $(find $O/pmc_write -name "*.db" | head -1) 12 64 $(find $O/pmc_tcc -name "*.db" | head -1) "$COMMIT" > $O/traffic.json 2>$O/traffic.err; echo "traffic rc=$?"; grep -E "fabric_bytes_per_frame\"|hit_rate" $O/traffic.json
# the traffic figure bench.py cites is the one just measured (profiles/r04_traffic.json carries its commit)
if [ -s $O/traffic.json ] && grep -q fabric_bytes_per_frame $O/traffic.json; then cp $O/traffic.json profiles/r04_traffic.json; fi
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-400 $O/bench_default.json; tail -2 $O/bench_default.err
C="--no-cpu-baseline --no-gray --shard-frames 0 --no-config-legs"
timeout 300 python bench.py --kind noise $C > $O/bench_noise.json 2> $O/bench_noise.err; echo "noise rc=$?"; cut -c1-160 $O/bench_noise.json
timeout 300 python bench.py --angle 0.8 $C > $O/bench_rot.json 2> $O/bench_rot.err; echo "rot rc=$?"; cut -c1-160 $O/bench_rot.json
timeout 300 python bench.py --angle 0.8 --face-rotation -79 $C > $O/bench_rot_rotfaces.json 2> $O/bench_rot_rotfaces.err; echo "rot_rotfaces rc=$?"; cut -c1-160 $O/bench_rot_rotfaces.json
timeout 300 python bench.py --rows 2160 --cols 3840 --min-size 20 --max-size 2000 --shift 0.05 --scale 1.05 --frames 8 --det-cap 32768 --gather-cap 64 --steps 5 --warmup 2 $C --verify-frames 2 > $O/bench_4k.json 2> $O/bench_4k.err; echo "4k rc=$?"; cut -c1-160 $O/bench_4k.json; tail -2 $O/bench_4k.err
timeout 300 python bench.py --frames 1 --steps 20 --warmup 5 $C --verify-frames 1 --no-single-frame > $O/bench_1frame.json 2> $O/bench_1frame.err; echo "1frame rc=$?"; cut -c1-300 $O/bench_1frame.json
python scripts/single_frame_latency.py 2>&1 | grep "single 1080p" | tee $O/single_frame.txt
# the C ABI's collective at world size 1 through a real RCCL communicator
timeout 300 python bench.py --force-dist --frames 32 --steps 3 --warmup 1 $C > $O/bench_dist1.json 2> $O/bench_dist1.err; echo "dist1 rc=$?"; cut -c1-160 $O/bench_dist1.json
# region phase timers (debug build): the small group alone and with the side chain next to it
[ -f pigo_amd/csrc/libpigo_hip_debug.so ] || python -m pigo_amd.build --debug > /dev/null 2>&1; (echo "## small group"; REG_ONLY=0 bash scripts/gpu_r4_phases.sh; echo "## mid group"; REG_ONLY=1 PHASE_SPECS="alone:PIGO_BIG_SKIP=3 with_side:PIGO_X=1" bash scripts/gpu_r4_phases.sh) > $O/region_phases.txt 2>&1; tail -8 $O/region_phases.txt
rm -rf $O/trace $O/trace_ov $O/pmc_fetch $O/pmc_write $O/pmc_tcc $O/pmc_sq $O/pmc_sq2
du -sh $O
