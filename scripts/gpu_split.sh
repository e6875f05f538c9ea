cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name="$1"; shift; env "$@" python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-gray --frames 128 $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'], d['kernel_ms'])"; }
run split160 PIGO_DEEP_SPLIT=160
run split192 PIGO_DEEP_SPLIT=192
run split224 PIGO_DEEP_SPLIT=224
run split256 PIGO_DEEP_SPLIT=256
run split320 PIGO_DEEP_SPLIT=320
run split440 PIGO_DEEP_SPLIT=440
EXTRA="--kind noise"
run noise128 X=1
run noise192 PIGO_DEEP_SPLIT=192
run noise256 PIGO_DEEP_SPLIT=256
EXTRA="--angle 0.8"
run rot128 X=1
run rot192 PIGO_DEEP_SPLIT=192
