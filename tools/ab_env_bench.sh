#!/bin/bash
# A/B of settings of environment knobs on the headline bench line (three frames in flight): tools/ab_env_bench.sh "A=1 B=2" "A=3" ...
set -u
cd "$GRAFT_REPO_ROOT"
for setting in "$@"; do
  ( for kv in $setting; do export "$kv"; done
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-exhaustive --no-other-arithmetic > gpurun_out/ab_env_bench.log 2>&1 || { echo "$setting failed"; tail -3 gpurun_out/ab_env_bench.log; exit 1; }
    python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/ab_env_bench.log') if l.startswith('{')][0])
print('$setting:', round(d['value'],1), 'Mpaths/s', round(d['ms_per_step'],4), 'ms/step; single', round(d.get('s_per_frame_single',0)*1e3,4))" ) || exit 1
done
