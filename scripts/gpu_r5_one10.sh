# Round 5, call 10: one-frame knobs -- chunk size (wave imbalance inside a region), hand-over tree, quad width
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5j; mkdir -p $O
export PIGO_TUNING=1
S=("one:" "c256:PIGO_REG_CHUNK0=256" "c128:PIGO_REG_CHUNK0=128" "c256_64:PIGO_REG_CHUNK0=256 PIGO_REG_CHUNK1=64" "c128_64:PIGO_REG_CHUNK0=128 PIGO_REG_CHUNK1=64" "c1_64:PIGO_REG_CHUNK1=64" "c1_256:PIGO_REG_CHUNK1=256"
   "nh38:PIGO_NH_LDS=38" "nh48:PIGO_NH_LDS=48" "nh20:PIGO_NH_LDS=20" "quad32:PIGO_REG_QUAD0=32 PIGO_REG_QUAD1=32" "quad16:PIGO_REG_QUAD0=16 PIGO_REG_QUAD1=16" "one_b:")
timeout 500 python scripts/ab.py --frames 1 --steps 100 --no-cluster --kernel-times "${S[@]}" 2>$O/ab_one.err | tee $O/ab_one.txt || tail -5 $O/ab_one.err
