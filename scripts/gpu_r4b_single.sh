# Round 4, second half, call 4: one 1080p frame per plan -- variant 2 (the default of pigo_run_cascade) against variant 3 under a few
# region settings, per-kernel times; then the debug library's phase timers of variant 3's groups on that frame.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; mkdir -p $O
L=$GRAFT_REPO_ROOT/pigo_amd/csrc
export PIGO_TUNING=1
S=("v2:" "v3:PIGO_SCAN_VARIANT=3" "v3_min128:PIGO_SCAN_VARIANT=3 PIGO_REG_MIN_REGIONS=128" "v3_min200:PIGO_SCAN_VARIANT=3 PIGO_REG_MIN_REGIONS=200"
   "v3_c256:PIGO_SCAN_VARIANT=3 PIGO_REG_CHUNK0=256" "v3_c128_64:PIGO_SCAN_VARIANT=3 PIGO_REG_CHUNK0=128 PIGO_REG_CHUNK1=64"
   "v3_c256_min200:PIGO_SCAN_VARIANT=3 PIGO_REG_MIN_REGIONS=200 PIGO_REG_CHUNK0=256" "v3_noquad:PIGO_SCAN_VARIANT=3 PIGO_REG_QUAD0=0 PIGO_REG_QUAD1=0"
   "v3_nobig:PIGO_SCAN_VARIANT=3 PIGO_BIG=0" "v2b:")
timeout 300 python scripts/ab_r4b.py --frames 1 --steps 50 --kernel-times "${S[@]}" 2>$O/ab_single.err | tee $O/ab_single.txt || tail -3 $O/ab_single.err
B="python bench.py --frames 1 --steps 20 --warmup 3 --no-cpu-baseline --no-gray --shard-frames 0 --no-single-frame --no-config-legs --verify-frames 0 --no-kernel-times"
for spec in "small:0" "mid:1"; do
  name="${spec%%:*}"; grp="${spec#*:}"
  echo "== $name"
  env PIGO_HIP_LIB=$L/libpigo_hip_debug.so PIGO_SCAN_VARIANT=3 PIGO_DEBUG_STATS=1 PIGO_REG_ONLY=$grp PIGO_BIG_SKIP=3 $B 2>&1 >/dev/null | grep "debug_stats raw" | python -c "
import sys,ast
for l in sys.stdin:
    st=ast.literal_eval(l.split('raw:')[1].strip())
    reg=max(st[4],1)
    print('regions %d | per region (cycles): copy %.0f scan %.0f wait %.0f deep(per wave) %.0f total %.0f | deep windows/region %.1f passes/window %.2f' % (st[4], st[0]/reg, st[1]/reg, st[3]/reg, st[2]/reg/16, st[5]/reg, st[7]/reg, st[6]/max(st[7],1)))
"
done 2>&1 | tee $O/phases_single.txt
