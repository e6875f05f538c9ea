# A/B of library builds (compile-time switches) on bench.py's default workload: one scripts/ab.py process per library, the same spec in each.
#   bash scripts/ab_libs.sh OUTFILE [--kernel-times ...] -- libpigo_hip.so libpigo_hip_x_foo.so ...
OUT=$1; shift
ARGS=""
while [ "$1" != "--" ] && [ $# -gt 0 ]; do ARGS="$ARGS $1"; shift; done
shift
: > $OUT
for L in "$@"; do
  env PIGO_HIP_LIB=$GRAFT_REPO_ROOT/pigo_amd/csrc/$L timeout 600 python scripts/ab.py $ARGS "$L:PIGO_X=1" >> $OUT 2>&1
done
grep -E "Gwin|MISMATCH|Error|error" $OUT
