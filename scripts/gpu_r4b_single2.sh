# one 1080p frame per plan: variant 3 without the side chain (PIGO_BIG=0: the big scales through the global tile class) under region settings
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; mkdir -p $O
export PIGO_TUNING=1
S=("v2:")
for m in 100 128 160 200 230; do for c0 in 256 512; do for c1 in 64 128; do
  S+=("v3_m${m}_c${c0}_${c1}:PIGO_SCAN_VARIANT=3 PIGO_BIG=0 PIGO_REG_MIN_REGIONS=$m PIGO_REG_CHUNK0=$c0 PIGO_REG_CHUNK1=$c1")
done; done; done
S+=("v3_m200_c256_128_s0_62:PIGO_SCAN_VARIANT=3 PIGO_BIG=0 PIGO_REG_MIN_REGIONS=200 PIGO_REG_CHUNK0=256 PIGO_REG_S0=62"
    "v3_m200_c256_128_s1_111:PIGO_SCAN_VARIANT=3 PIGO_BIG=0 PIGO_REG_MIN_REGIONS=200 PIGO_REG_CHUNK0=256 PIGO_REG_S1=111"
    "v3_m200_c256_128_s1_90:PIGO_SCAN_VARIANT=3 PIGO_BIG=0 PIGO_REG_MIN_REGIONS=200 PIGO_REG_CHUNK0=256 PIGO_REG_S1=90"
    "v3_m200_c256_deep:PIGO_SCAN_VARIANT=3 PIGO_BIG=0 PIGO_REG_MIN_REGIONS=200 PIGO_REG_CHUNK0=256 PIGO_REG_DEEP_SMALL=1" "v2b:")
timeout 300 python scripts/ab_r4b.py --frames 1 --steps 50 "${S[@]}" 2>$O/ab_single2.err | tee $O/ab_single2.txt || tail -3 $O/ab_single2.err
T=("v2:" "v3_m200_c256:PIGO_SCAN_VARIANT=3 PIGO_BIG=0 PIGO_REG_MIN_REGIONS=200 PIGO_REG_CHUNK0=256" "v3_m128_c256:PIGO_SCAN_VARIANT=3 PIGO_BIG=0 PIGO_REG_MIN_REGIONS=128 PIGO_REG_CHUNK0=256")
timeout 300 python scripts/ab_r4b.py --frames 1 --steps 50 --kernel-times "${T[@]}" 2>>$O/ab_single2.err | tee -a $O/ab_single2.txt
