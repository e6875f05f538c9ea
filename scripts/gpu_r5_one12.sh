# Round 5, call 12: one-launch plans for every small plan whose region groups fit (grid search honours the packed-offset limit)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5l; mkdir -p $O
python scripts/dbg_sizes.py 2>&1 | grep -v "^\[pigo\] \(sync\|launch\)" | grep -v "^\[pigo\] k_scan_one:" | tail -30 | tee $O/sizes.txt
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 | tee $O/pytest.txt
timeout 300 python scripts/one_stress.py --launches 1500 --sizes 480x640,1080x1920 --frames 7 2>$O/stress.err | tee $O/stress.txt || tail -5 $O/stress.err
