cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
t() { echo "== $*"; env "$@" 2>&1 | grep -v amdgpu.ids | tail -3; }
t X=1 python scripts/gpu_r2_dbg.py faces
t X=1 python scripts/gpu_r2_dbg.py noise
t X=1 python scripts/gpu_r2_dbg.py sample 20 1000 0.1
t X=1 python scripts/gpu_r2_dbg.py sample 20 1000 0.2
t PIGO_SCAN_VARIANT=2 python scripts/gpu_r2_dbg.py faces
