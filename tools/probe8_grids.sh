#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for setting in "X=1" "RT_AMD_FUSED_GRID=512" "RT_AMD_FUSED_GRID=256" "RT_AMD_BIN_GRID=512" "RT_AMD_BIN_GRID=256 RT_AMD_FUSED_GRID=256" "RT_AMD_BIN_GRID=512 RT_AMD_FUSED_GRID=384"; do
  ( for kv in $setting; do export "$kv"; done; echo "## $setting"; timeout -k 10 120 python3 tools/scaling_probe.py 8 three 2>&1 | grep "^share" ) || exit 1
done
