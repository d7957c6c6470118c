#!/bin/bash
# A/B of library builds on the headline bench line (three frames in flight): tools/ab_bench.sh NAME...   (NAME = base, or a variant of tools/build_variant.sh)
set -u
cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  if [ "$v" = base ]; then unset RT_AMD_LIB; else export RT_AMD_LIB=$GRAFT_REPO_ROOT/raytracer_amd/variants/librt_amd_$v.so; fi
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-exhaustive --no-other-arithmetic ${BENCH_ARGS:-} > gpurun_out/ab_bench_$v.log 2>&1 || { echo "$v failed"; tail -3 gpurun_out/ab_bench_$v.log; exit 1; }
  python3 -c "
import json,sys
d=json.loads([l for l in open('gpurun_out/ab_bench_$v.log') if l.startswith('{')][0])
print('$v', round(d['value'],1), 'Mpaths/s', round(d['ms_per_step'],4), 'ms/step; single', round(d.get('s_per_frame_single',0)*1e3,4))"
done
