#!/bin/bash
# A/B of library builds on a rank's share of the headline frame (tools/scaling_probe.py): tools/ab_share.sh "probe words" NAME...
set -u
cd "$GRAFT_REPO_ROOT"
WORDS=${1:?probe words, e.g. "8 three serial fast"}; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset RT_AMD_LIB; else export RT_AMD_LIB=$GRAFT_REPO_ROOT/raytracer_amd/variants/librt_amd_$v.so; fi
  echo "== $v"
  timeout -k 10 200 python3 tools/scaling_probe.py $WORDS 2>&1 | grep "^share" || exit 1
done
