#!/bin/bash
# rocprofv3 kernel-trace summary of the bench: tools/prof_kernels.sh <out-dir> [bench args]
set -u
OUT=${1:?out dir}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$OUT.log" 2>&1 || { echo "profile run failed"; tail -5 "$OUT.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/*/*kernel_stats.csv'):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(r['Name'].replace('(anonymous namespace)::', '')[:30].ljust(32), r['Calls'].rjust(4),
              '%10.3f ms total %9.3f ms avg %5.1f%%' % (float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e6, float(r['Percentage'])))
PY
grep -v "^[EW]2026" "$OUT.log" | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'Mpaths/s', d['value'])"
