cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  name="${spec%%:*}"; envs="${spec#*:}"
  echo "== $name"; env $envs python scripts/single_frame_latency.py 2>&1 | grep "single 1080p"
done
