# Round 4, second half, call 2: the quad pass of the region kernel's deep list (PIGO_REG_QUAD0/1) -- parity, then A/B on the default
# workload, the rotated config (both face sets) and the 4K config, then the phase timers with and without it.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; mkdir -p $O
L=$GRAFT_REPO_ROOT/pigo_amd/csrc
export PIGO_TUNING=1
timeout 900 python -m pytest tests -m gpu -q -x -k "${PYTEST_K:-quad or deep_list or golden or random_parameter_sweep}" > $O/pytest_quad.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_quad.log | tail -2 | cut -c1-300
S=("def:" "q1_16:PIGO_REG_QUAD1=16" "q0_32:PIGO_REG_QUAD0=32" "q0_16:PIGO_REG_QUAD0=16" "q0_32_q1_16:PIGO_REG_QUAD0=32 PIGO_REG_QUAD1=16"
   "q1_32:PIGO_REG_QUAD1=32" "q1_16_nh9:PIGO_REG_QUAD1=16 PIGO_NH_REG1=9" "q1_16_nh18:PIGO_REG_QUAD1=16 PIGO_NH_REG1=18"
   "q0_32_nh18:PIGO_REG_QUAD0=32 PIGO_NH_LDS=18" "def2:")
timeout 600 python scripts/ab_r4b.py --kernel-times "${S[@]}" 2>$O/ab_quad.err | tee $O/ab_quad.txt || tail -3 $O/ab_quad.err
R=("def:" "q1_16:PIGO_REG_QUAD1=16" "q0_32_q1_16:PIGO_REG_QUAD0=32 PIGO_REG_QUAD1=16")
timeout 300 python scripts/ab_r4b.py --frames 64 --angle 0.8 "${R[@]}" 2>$O/ab_quad_rot.err | tee $O/ab_quad_rot.txt
timeout 300 python scripts/ab_r4b.py --frames 64 --angle 0.8 --face-rotation -79 "${R[@]}" 2>$O/ab_quad_rotf.err | tee $O/ab_quad_rotf.txt
K="--rows 2160 --cols 3840 --min-size 20 --max-size 2000 --shift 0.05 --scale 1.05 --frames 8 --det-cap 32768 --steps 5 --reps 2"
timeout 300 python scripts/ab_r4b.py $K "${R[@]}" 2>$O/ab_quad_4k.err | tee $O/ab_quad_4k.txt
if [ -f $L/libpigo_hip_debug.so ]; then
  B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-gray --shard-frames 0 --no-single-frame --no-config-legs --verify-frames 0 --no-kernel-times"
  for spec in "small:0:PIGO_X=1" "small_q32:0:PIGO_REG_QUAD0=32" "mid:1:PIGO_X=1" "mid_q16:1:PIGO_REG_QUAD1=16"; do
    name="${spec%%:*}"; rest="${spec#*:}"; grp="${rest%%:*}"; envs="${rest#*:}"
    echo "== $name"
    env PIGO_HIP_LIB=$L/libpigo_hip_debug.so PIGO_DEBUG_STATS=1 PIGO_REG_ONLY=$grp PIGO_BIG_SKIP=3 $envs $B 2>&1 >/dev/null | grep "debug_stats raw" | python -c "
import sys,ast
for l in sys.stdin:
    st=ast.literal_eval(l.split('raw:')[1].strip())
    reg=max(st[4],1)
    print('regions %d | per region (cycles): copy %.0f scan %.0f wait %.0f deep(per wave) %.0f total %.0f | deep windows/region %.1f passes/window %.2f' % (st[4], st[0]/reg, st[1]/reg, st[3]/reg, st[2]/reg/16, st[5]/reg, st[7]/reg, st[6]/max(st[7],1)))
"
  done 2>&1 | tee $O/phases_quad.txt
fi
