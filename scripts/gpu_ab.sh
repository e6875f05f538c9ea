cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/pigo_amd/csrc/libpigo_hip_ptrack.so
PIGO_HIP_LIB=$V timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2 | cut -c1-300
run() { name="$1"; shift; env "$@" python bench.py --frames 64 --steps 8 --warmup 2 --no-cpu-baseline --no-gray $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['kernel_ms'])"; }
run base X=1
run ptrack PIGO_HIP_LIB=$V
run base X=1
run ptrack PIGO_HIP_LIB=$V
EXTRA="--angle 0.8"
run rot_base X=1
run rot_ptrack PIGO_HIP_LIB=$V
EXTRA="--kind noise"
run noise_base X=1
run noise_ptrack PIGO_HIP_LIB=$V
