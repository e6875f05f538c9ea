cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PIGO_TUNING=1
for spec in "default:PIGO_X=1" "batch_prio0:PIGO_SIDE_PRIO=0" "noprobe:PIGO_SIDE_PROBE=0" "prio0_noprobe:PIGO_SIDE_PRIO=0 PIGO_SIDE_PROBE=0"; do
  name="${spec%%:*}"; envs="${spec#*:}"
  echo "== $name"
  env $envs timeout 200 python scripts/single_in_process.py 3 2>&1 | grep "one-frame"
done
