# Round 5, call 20: the side chain walks the batch in chunks of 64 frames (PIGO_BIG_CHUNK_FRAMES) -- 128 / 512 / 1,024 frames per step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5u; mkdir -p $O
export PIGO_TUNING=1
for n in 128 512 1024; do
  timeout 400 python scripts/ab.py --frames $n --steps 4 --reps 2 --kernel-times "chunk64_$n:" "whole_$n:PIGO_BIG_CHUNK_FRAMES=0" "chunk32_$n:PIGO_BIG_CHUNK_FRAMES=32" "chunk128_$n:PIGO_BIG_CHUNK_FRAMES=128" 2>$O/ab.err | tee -a $O/ab.txt || tail -5 $O/ab.err
done
