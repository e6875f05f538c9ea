# Round 5, call 7: the last workgroup clears the poison nobody took -- hand-off stress again
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5g; mkdir -p $O
timeout 500 python scripts/one_stress.py --launches 4000 2>$O/stress.err | tee $O/stress.txt || tail -5 $O/stress.err
timeout 200 python scripts/one_stress.py --launches 2000 --sizes 1080x1920 --frames 3 2>>$O/stress.err | tee -a $O/stress.txt
timeout 200 python scripts/one_stress.py --launches 2000 --sizes 1080x1920 --angle 0.8 2>>$O/stress.err | tee -a $O/stress.txt
timeout 200 python scripts/one_stress.py --launches 2000 --sizes 1080x1920 --kind noise 2>>$O/stress.err | tee -a $O/stress.txt
