#!/bin/bash
# A/B of settings of several environment knobs on the headline frame: tools/ab_env2.sh "A=1 B=2" "A=3" ...
set -u
cd "$GRAFT_REPO_ROOT"
i=0
for setting in "$@"; do
  i=$((i+1))
  ( for kv in $setting; do export "$kv"; done; echo "## $setting"; bash tools/prof_frame.sh env$i ) || exit 1
done
