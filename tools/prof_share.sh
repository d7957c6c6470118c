#!/bin/bash
# rocprofv3 kernel summary of the per-rank load of an N-GPU run: tools/prof_share.sh <out-dir> <N>
set -u
OUT=${1:?out dir}; N=${2:-8}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 tools/scaling_probe.py "$N" > "$OUT.log" 2>&1 || { echo "profile run failed"; tail -5 "$OUT.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/*/*kernel_stats.csv'):
    rows = list(csv.DictReader(open(f)))
    total = sum(float(r['TotalDurationNs']) for r in rows)
    for r in rows[:12]:
        print(r['Name'].replace('(anonymous namespace)::', '')[:34].ljust(36), r['Calls'].rjust(5),
              '%10.3f ms total %9.4f ms avg %5.1f%%' % (float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e6, float(r['Percentage'])))
    print('all kernels: %.3f ms' % (total / 1e6))
PY
grep share "$OUT.log"
