# how hard the side chain's tail leans on the region kernel: waves per CU of k_tail_deep next to the region launches
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; mkdir -p $O
export PIGO_TUNING=1
S=("def:" "tt384:PIGO_BIG_TAIL_THREADS=384" "tt256:PIGO_BIG_TAIL_THREADS=256" "tt128:PIGO_BIG_TAIL_THREADS=128" "pw2:PIGO_BIG_POOL_WAVES=2" "nh28:PIGO_NH_BIG=28" "nh6:PIGO_NH_BIG=6" "def2:")
timeout 300 python scripts/ab_r4b.py "${S[@]}" 2>$O/ab_tail.err | tee $O/ab_tail.txt || tail -3 $O/ab_tail.err
K="--rows 2160 --cols 3840 --min-size 20 --max-size 2000 --shift 0.05 --scale 1.05 --frames 8 --det-cap 32768 --steps 5 --reps 2"
timeout 300 python scripts/ab_r4b.py $K "def:" "tt384:PIGO_BIG_TAIL_THREADS=384" "tt256:PIGO_BIG_TAIL_THREADS=256" 2>$O/ab_tail_4k.err | tee $O/ab_tail_4k.txt
