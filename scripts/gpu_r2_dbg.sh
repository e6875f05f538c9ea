cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
t() { echo "== $*"; env "$@" 2>&1 | grep -v amdgpu.ids | tail -2; }
t X=1 python scripts/gpu_r2_dbg.py sample 20 1000 0.1
t X=1 python scripts/gpu_r2_dbg.py sample 20 48 0.2
t X=1 python scripts/gpu_r2_dbg.py sample 50 140 0.2
t X=1 python scripts/gpu_r2_dbg.py sample 150 1000 0.2
t X=1 python scripts/gpu_r2_dbg.py sample 150 250 0.2
t X=1 python scripts/gpu_r2_dbg.py sample 250 1000 0.2
t X=1 python scripts/gpu_r2_dbg.py sample 250 1000 0.1
