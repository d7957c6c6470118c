#!/bin/bash
# A/B of one environment knob on the headline frame: tools/ab_env.sh VAR value...
set -u
cd "$GRAFT_REPO_ROOT"
VAR=${1:?variable}; shift
for v in "$@"; do
  export $VAR=$v
  bash tools/prof_frame.sh ${VAR}_$v || exit 1
done
