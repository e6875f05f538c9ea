# Round 5, call 19: k_tail_deep of the side chain takes its entries in queue order from a counter (PIGO_BIG_TAIL_CLAIM) -- 128 / 512 / 1,024 frames per step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5t; mkdir -p $O
export PIGO_TUNING=1
for n in 128 512 1024; do
  timeout 400 python scripts/ab.py --frames $n --steps 4 --reps 2 --kernel-times "claim$n:" "static$n:PIGO_BIG_TAIL_CLAIM=0" "claim${n}_b:" 2>$O/ab.err | tee -a $O/ab.txt || tail -5 $O/ab.err
done
timeout 300 python scripts/ab.py --rows 2160 --cols 3840 --min-size 20 --max-size 2000 --shift 0.05 --scale 1.05 --frames 8 --det-cap 32768 --steps 4 --reps 2 --kernel-times "claim4k:" "static4k:PIGO_BIG_TAIL_CLAIM=0" 2>$O/ab4k.err | tee -a $O/ab.txt || tail -5 $O/ab4k.err
