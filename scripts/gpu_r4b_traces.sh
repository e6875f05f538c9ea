# kernel traces (rocprofv3 --kernel-trace --stats) of the 4K config and of the rotated config on rotated faces: the summaries go to profiles/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; mkdir -p $O
COMMIT="${COMMIT:-unknown}"
C="--no-cpu-baseline --no-gray --shard-frames 0 --no-config-legs --no-single-frame --verify-frames 0"
K="python bench.py --rows 2160 --cols 3840 --min-size 20 --max-size 2000 --shift 0.05 --scale 1.05 --frames 8 --det-cap 32768 --gather-cap 64 --steps 5 --warmup 2 $C"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/tr4k -o t -- $K > $O/tr4k.log 2>&1; echo "trace4k rc=$?"
python scripts/summarize_prof.py "round 4 (scripts/gpu_r4b_traces.sh, commit $COMMIT): $K -- 8 x 4K frames per step, 12 scan steps (2 warm-up + 5 timed + 5 per-kernel event reps, the last five with every launch alone on one stream)" $(find $O/tr4k -name "*.db" | head -1) > $O/trace4k_summary.txt 2>$O/trace4k_summary.err; head -16 $O/trace4k_summary.txt | cut -c1-120
R="python bench.py --angle 0.8 --face-rotation -79 --frames 64 --steps 5 --warmup 2 $C"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trrot -o t -- $R > $O/trrot.log 2>&1; echo "tracerot rc=$?"
python scripts/summarize_prof.py "round 4 (scripts/gpu_r4b_traces.sh, commit $COMMIT): $R -- 64 x 1080p frames with rotated faces per step, angle 0.8, 12 scan steps as above" $(find $O/trrot -name "*.db" | head -1) > $O/tracerot_summary.txt 2>$O/tracerot_summary.err; head -12 $O/tracerot_summary.txt | cut -c1-120
rm -rf $O/tr4k $O/trrot
