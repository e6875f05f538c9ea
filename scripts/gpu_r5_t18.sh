# Round 5, call 18: k_scan_one -- the EXCESS of a long deep list (beyond PIGO_ONE_KEEP entries) goes to the idle consumers
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s; mkdir -p $O
export PIGO_TUNING=1
S=("def:" "off0:PIGO_ONE_OFFLOAD_X10=0" "k64:PIGO_ONE_KEEP=64" "k128:PIGO_ONE_KEEP=128" "k160:PIGO_ONE_KEEP=160" "k32:PIGO_ONE_KEEP=32" "x10:PIGO_ONE_OFFLOAD_X10=10" "x40:PIGO_ONE_OFFLOAD_X10=40" "x80:PIGO_ONE_OFFLOAD_X10=80" "k64x40:PIGO_ONE_KEEP=64 PIGO_ONE_OFFLOAD_X10=40" "k128_ntl2:PIGO_ONE_KEEP=128 PIGO_ONE_NT_LATE=2" "k96_nt2:PIGO_ONE_NT=2" "off0_b:PIGO_ONE_OFFLOAD_X10=0" "def_b:")
timeout 500 python scripts/ab.py --frames 1 --steps 100 --no-cluster "${S[@]}" 2>$O/ab_one.err | tee $O/ab_one.txt || tail -5 $O/ab_one.err
timeout 200 python scripts/ab.py --frames 1 --steps 100 --no-cluster --kind noise "def:" "off0:PIGO_ONE_OFFLOAD_X10=0" "k128:PIGO_ONE_KEEP=128" "k160:PIGO_ONE_KEEP=160" 2>$O/ab_one_noise.err | tee $O/ab_one_noise.txt || tail -5 $O/ab_one_noise.err
timeout 200 python scripts/ab.py --frames 1 --steps 100 --no-cluster --angle 0.8 "def:" "off0:PIGO_ONE_OFFLOAD_X10=0" 2>$O/ab_one_rot.err | tee $O/ab_one_rot.txt || tail -5 $O/ab_one_rot.err
timeout 200 python scripts/ab.py --frames 1 --steps 100 --no-cluster --rows 720 --cols 1280 "def:" "off0:PIGO_ONE_OFFLOAD_X10=0" 2>$O/ab_one_720.err | tee $O/ab_one_720.txt || tail -5 $O/ab_one_720.err
timeout 200 python scripts/ab.py --frames 3 --steps 50 --no-cluster "def:" "off0:PIGO_ONE_OFFLOAD_X10=0" 2>$O/ab_three.err | tee $O/ab_three.txt || tail -5 $O/ab_three.err
