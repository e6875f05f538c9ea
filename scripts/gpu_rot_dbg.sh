cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/pytest_rot.log 2>&1; echo rc=$?
grep -n "Error\|error\|FAILED\|passed\|failed" gpurun_out/pytest_rot.log | head -30
echo ---- serialized
AMD_SERIALIZE_KERNEL=3 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/pytest_rot_ser.log 2>&1; echo rc=$?
grep -n "Error\|error\|FAILED\|passed\|failed" gpurun_out/pytest_rot_ser.log | head -30
echo ---- rot lds off
PIGO_ROT_LDS=0 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
