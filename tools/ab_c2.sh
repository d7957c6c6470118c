#!/bin/bash
# A/B of library builds on the sphere-only configurations (C1, C2, a 100-sample C5): tools/ab_c2.sh NAME...   (NAME = base, or a variant of tools/build_variant.sh)
set -u
cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  if [ "$v" = base ]; then unset RT_AMD_LIB; else export RT_AMD_LIB=$GRAFT_REPO_ROOT/raytracer_amd/variants/librt_amd_$v.so; fi
  echo "== $v"
  timeout -k 10 200 python3 tools/bench_configs.py --arithmetic 1 --no-exhaustive --only default --c5-spp 100 2>&1 | grep -E "^\| (C1|C2|C5)" || exit 1
done
