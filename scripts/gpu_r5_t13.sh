# Round 5, call 13: wire flags word, comm locks, deadline subprocess, the suite without PIGO_TUNING
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5n; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 | tee $O/pytest.txt
