cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest "tests/test_gpu_parity.py::test_chunked_pipeline_large_batches" -x -q 2>&1 | grep -v "^$" | grep "Error\|error\|assert\|FAILED\|passed\|failed\|E  " | head -30
echo --- rot qb1
PIGO_ROT_QB_DIV=1 timeout 600 python -m pytest "tests/test_gpu_parity.py::test_chunked_pipeline_large_batches" -x -q 2>&1 | tail -2
echo --- rot no lds
PIGO_ROT_LDS=0 timeout 600 python -m pytest "tests/test_gpu_parity.py::test_chunked_pipeline_large_batches" -x -q 2>&1 | tail -2
