# phase timers of the small region group (debug build) with and without the side chain next to it
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PIGO_TUNING=1  # the settings below are tuning switches (ignored without it)
mkdir -p gpurun_out/r4
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-gray --shard-frames 0 --no-single-frame --no-config-legs --verify-frames 0 --no-kernel-times"
for spec in ${PHASE_SPECS:-alone:PIGO_BIG_SKIP=3 with_side:PIGO_X=1 with_big_only:PIGO_BIG_SKIP=1 with_tail_only:PIGO_BIG_SKIP=2} ${EXTRA:-}; do
  name="${spec%%:*}"; envs="${spec#*:}"
  echo "== $name"
  env PIGO_HIP_LIB=$GRAFT_REPO_ROOT/pigo_amd/csrc/libpigo_hip_debug.so PIGO_DEBUG_STATS=1 PIGO_REG_ONLY=${REG_ONLY:-0} $envs $B 2>&1 >/dev/null | grep "debug_stats raw" | python -c "
import sys,ast
for l in sys.stdin:
    st=ast.literal_eval(l.split('raw:')[1].strip())
    reg=max(st[4],1)
    print('regions %d | per region (cycles): copy %.0f scan %.0f wait %.0f deep(per wave) %.0f total %.0f | deep windows/region %.1f passes/window %.2f' % (st[4], st[0]/reg, st[1]/reg, st[3]/reg, st[2]/reg/16, st[5]/reg, st[7]/reg, st[6]/max(st[7],1)))
"
done
