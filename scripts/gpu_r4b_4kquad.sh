cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; mkdir -p $O
export PIGO_TUNING=1
K="--rows 2160 --cols 3840 --min-size 20 --max-size 2000 --shift 0.05 --scale 1.05 --frames 8 --det-cap 32768 --steps 5 --reps 3 --kernel-times"
timeout 200 python scripts/ab_r4b.py $K "def:" "q1_0:PIGO_REG_QUAD1=0" "q1_32:PIGO_REG_QUAD1=32" "def2:" "q1_0b:PIGO_REG_QUAD1=0" 2>$O/ab_4kquad.err | tee $O/ab_4kquad.txt
