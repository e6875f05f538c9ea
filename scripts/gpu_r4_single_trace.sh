# kernel timeline of single-frame calls: bash scripts/gpu_r4_single_trace.sh NAME "ENV=.."
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PIGO_TUNING=1  # the settings below are tuning switches (ignored without it)
mkdir -p gpurun_out/r4
name="$1"; envs="$2"
echo -n "$name: "; env $envs python scripts/single_frame_latency.py 2>&1 | grep "single 1080p" | sed 's/single 1080p frame: //' | cut -c1-200
rm -rf /tmp/trs_$name
timeout 300 env $envs rocprofv3 --kernel-trace -d /tmp/trs_$name -o t -- python scripts/single_frame_latency.py > gpurun_out/r4/strace_$name.log 2>&1
db=$(find /tmp/trs_$name -name "*.db" | head -1)
python scripts/trace_overlap.py "$db" 1 > gpurun_out/r4/strace_$name.txt 2>&1; grep -v "^#" gpurun_out/r4/strace_$name.txt | head -24
