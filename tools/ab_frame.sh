#!/bin/bash
# A/B of library builds on the headline frame: tools/ab_frame.sh NAME...   (NAME = base, or a variant of tools/build_variant.sh)
set -u
cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  if [ "$v" = base ]; then unset RT_AMD_LIB; else export RT_AMD_LIB=$GRAFT_REPO_ROOT/raytracer_amd/variants/librt_amd_$v.so; fi
  bash tools/prof_frame.sh $v || exit 1
done
