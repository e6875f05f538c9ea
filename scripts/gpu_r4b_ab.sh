# Round 4, second half: one GPU call that (1) runs the parity suite on the library with every PIGO_OPT_* switch on, (2) times the
# compile-time variants (scripts/build_r4b_variants.py built them) and (3) the run-time schedule switches on bench.py's default
# workload, every setting oracle-verified (scripts/ab_r4b.py); (4) rotated and 4K legs; (5) phase timers of the mid group.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; mkdir -p $O
L=$GRAFT_REPO_ROOT/pigo_amd/csrc
export PIGO_TUNING=1
ALL=$L/libpigo_hip_x_all.so
PIGO_HIP_LIB=$ALL timeout 900 python -m pytest tests -m gpu -q -x ${PYTEST_ARGS:-} > $O/pytest_all.log 2>&1; echo "pytest(all) rc=$?"; tail -3 $O/pytest_all.log | cut -c1-300
for v in base all dec lm l1 cx nol1; do
  PIGO_HIP_LIB=$L/libpigo_hip_x_$v.so timeout 300 python scripts/ab_r4b.py --kernel-times "$v:" 2>$O/ab_$v.err | tee -a $O/ab_libs.txt | grep -v "^#" || tail -3 $O/ab_$v.err
done
S=(
 "def:"
 "t0_256:PIGO_REG_TAPER0=256"
 "t0_128:PIGO_REG_TAPER0=128"
 "t0_128m1:PIGO_REG_TAPER0=128 PIGO_REG_TAPER_MUL0=1"
 "t0_128m4:PIGO_REG_TAPER0=128 PIGO_REG_TAPER_MUL0=4"
 "t0_64:PIGO_REG_TAPER0=64"
 "t1_64:PIGO_REG_TAPER1=64"
 "t1_64m1:PIGO_REG_TAPER1=64 PIGO_REG_TAPER_MUL1=1"
 "t1_64m4:PIGO_REG_TAPER1=64 PIGO_REG_TAPER_MUL1=4"
 "c1_256_t64:PIGO_REG_CHUNK1=256 PIGO_REG_TAPER1=64 PIGO_REG_TAPER_MUL1=4"
 "t01:PIGO_REG_TAPER0=128 PIGO_REG_TAPER1=64"
 "merge:PIGO_REG_MERGE_LAUNCH=1"
 "merge_t01:PIGO_REG_MERGE_LAUNCH=1 PIGO_REG_TAPER0=128 PIGO_REG_TAPER1=64"
 "res1_8:PIGO_REG_RESERVE1_KB=8"
 "par:PIGO_REG_PAR=1"
 "par_t01:PIGO_REG_PAR=1 PIGO_REG_TAPER0=128 PIGO_REG_TAPER1=64"
 "s0_57:PIGO_REG_S0=57"
 "s1_163:PIGO_REG_S1=163"
 "s1_135:PIGO_REG_S1=135"
 "def2:"
)
PIGO_HIP_LIB=$ALL timeout 600 python scripts/ab_r4b.py "${S[@]}" 2>$O/ab_sched.err | tee $O/ab_sched.txt || tail -3 $O/ab_sched.err
# rotated (config 4) and the 4K stress config (config 5): base vs all vs all + the schedule switches
R=("def:" "t01:PIGO_REG_TAPER0=128 PIGO_REG_TAPER1=64" "merge_t01:PIGO_REG_MERGE_LAUNCH=1 PIGO_REG_TAPER0=128 PIGO_REG_TAPER1=64")
PIGO_HIP_LIB=$L/libpigo_hip_x_base.so timeout 300 python scripts/ab_r4b.py --frames 64 --angle 0.8 "base:" 2>$O/ab_rot_base.err | tee $O/ab_rot.txt
PIGO_HIP_LIB=$ALL timeout 300 python scripts/ab_r4b.py --frames 64 --angle 0.8 "${R[@]}" 2>$O/ab_rot_all.err | tee -a $O/ab_rot.txt
K="--rows 2160 --cols 3840 --min-size 20 --max-size 2000 --shift 0.05 --scale 1.05 --frames 8 --det-cap 32768 --steps 5 --reps 2"
PIGO_HIP_LIB=$L/libpigo_hip_x_base.so timeout 300 python scripts/ab_r4b.py $K "base:" 2>$O/ab_4k_base.err | tee $O/ab_4k.txt
PIGO_HIP_LIB=$ALL timeout 300 python scripts/ab_r4b.py $K "${R[@]}" 2>$O/ab_4k_all.err | tee -a $O/ab_4k.txt
# phase timers (debug build of the "all" setting; scripts/build_r4b_variants.py debug): each group alone, with and without the taper
if [ -f $L/libpigo_hip_debug.so ]; then
  B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-gray --shard-frames 0 --no-single-frame --no-config-legs --verify-frames 0 --no-kernel-times"
  for spec in "small:0:PIGO_X=1" "small_t128:0:PIGO_REG_TAPER0=128" "mid:1:PIGO_X=1" "mid_t64:1:PIGO_REG_TAPER1=64"; do
    name="${spec%%:*}"; rest="${spec#*:}"; grp="${rest%%:*}"; envs="${rest#*:}"
    echo "== $name"
    env PIGO_HIP_LIB=$L/libpigo_hip_debug.so PIGO_DEBUG_STATS=1 PIGO_REG_ONLY=$grp PIGO_BIG_SKIP=3 $envs $B 2>&1 >/dev/null | grep "debug_stats raw" | python -c "
import sys,ast
for l in sys.stdin:
    st=ast.literal_eval(l.split('raw:')[1].strip())
    reg=max(st[4],1)
    print('regions %d | per region (cycles): copy %.0f scan %.0f wait %.0f deep(per wave) %.0f total %.0f | deep windows/region %.1f passes/window %.2f' % (st[4], st[0]/reg, st[1]/reg, st[3]/reg, st[2]/reg/16, st[5]/reg, st[7]/reg, st[6]/max(st[7],1)))
"
  done 2>&1 | tee $O/phases.txt
fi
