# kernel-trace of the default overlapped step:  bash scripts/gpu_r4_trace.sh NAME "ENV=.."   -> gpurun_out/r4/trace_NAME.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PIGO_TUNING=1  # the settings below are tuning switches (ignored without it)
mkdir -p gpurun_out/r4
name="$1"; envs="$2"
B="python bench.py --frames ${FRAMES:-128} --steps 6 --warmup 2 --no-cpu-baseline --no-gray --no-single-frame --shard-frames 0 --verify-frames 0 --no-kernel-times ${BENCH_ARGS:-}"
rm -rf /tmp/tr_$name
timeout 600 env $envs rocprofv3 --kernel-trace --stats -d /tmp/tr_$name -o t -- $B > gpurun_out/r4/trace_$name.log 2>&1; echo "trace rc=$?"
db=$(find /tmp/tr_$name -name "*.db" | head -1)
python scripts/trace_overlap.py "$db" 1 > gpurun_out/r4/trace_$name.txt 2>&1; tail -25 gpurun_out/r4/trace_$name.txt
