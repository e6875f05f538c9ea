# per-group phase timers of k_scan_region (debug build): copy / wave-0 scan / deep cycles per region
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
for g in 0 1; do
  env PIGO_HIP_LIB=$GRAFT_REPO_ROOT/pigo_amd/csrc/libpigo_hip_debug.so PIGO_DEBUG_STATS=1 PIGO_REG_ONLY=$g python bench.py --frames 64 --steps 2 --warmup 1 --no-cpu-baseline --no-gray --shard-frames 0 --verify-frames 0 --no-single-frame > gpurun_out/r2/regdbg_$g.json 2> gpurun_out/r2/regdbg_$g.err
  echo "group $g"; grep "debug_stats raw" gpurun_out/r2/regdbg_$g.err | tail -1
done
