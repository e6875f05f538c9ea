# Round 4, second half, call 3: quad pass on by default (32 / 16) with its leaves staged in LDS: parity subset, A/B against the global leaves
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; mkdir -p $O
export PIGO_TUNING=1
timeout 900 python -m pytest tests -m gpu -q -x -k "${PYTEST_K:-quad or deep_list or golden or random_parameter_sweep or rotated_region}" > $O/pytest_leaf.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_leaf.log | tail -2 | cut -c1-300
S=("def:" "leaf0:PIGO_REG_QUAD_LEAF=0" "q1_32:PIGO_REG_QUAD1=32" "noquad:PIGO_REG_QUAD0=0 PIGO_REG_QUAD1=0" "prio0:PIGO_REG_PRIO=0" "prio2:PIGO_REG_PRIO=2" "wq512:PIGO_REG_WQ=512" "def2:")
timeout 600 python scripts/ab_r4b.py --kernel-times "${S[@]}" 2>$O/ab_leaf.err | tee $O/ab_leaf.txt || tail -3 $O/ab_leaf.err
R=("def:" "leaf0:PIGO_REG_QUAD_LEAF=0")
timeout 300 python scripts/ab_r4b.py --frames 64 --angle 0.8 --face-rotation -79 "${R[@]}" 2>$O/ab_leaf_rotf.err | tee $O/ab_leaf_rotf.txt
