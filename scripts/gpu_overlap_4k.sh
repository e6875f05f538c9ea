# timeline of one timed 4K step (BASELINE configs[4]): bash scripts/gpu_overlap_4k.sh OUTDIR
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$1; mkdir -p $O
T="python bench.py --rows 2160 --cols 3840 --min-size 20 --max-size 2000 --shift 0.05 --scale 1.05 --frames 8 --det-cap 32768 --gather-cap 64 --steps 4 --warmup 2 --no-cpu-baseline --no-gray --no-single-frame --no-config-legs --shard-frames 0 --verify-frames 0 --no-kernel-times"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_4k -o t -- $T > $O/trace_4k.log 2>&1; echo "trace rc=$?"
python scripts/trace_overlap.py $(find $O/trace_4k -name "*.db" | head -1) 1 > $O/overlap_4k.txt 2>&1
python scripts/summarize_prof.py "4K step (scripts/gpu_overlap_4k.sh): $T" $(find $O/trace_4k -name "*.db" | head -1) > $O/trace4k_summary.txt 2>/dev/null
rm -rf $O/trace_4k
tail -30 $O/overlap_4k.txt
