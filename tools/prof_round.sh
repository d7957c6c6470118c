#!/bin/bash
# Kernel-trace summaries of the bench for profiles/: tools/prof_round.sh <tag>
# (one frame at a time, three frames in flight, one frame launch by launch, per-rank shares)
set -u
TAG=${1:?tag}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG}
mkdir -p $OUT
summ() {
python3 - "$1" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/*/*kernel_stats.csv'):
    for r in list(csv.DictReader(open(f)))[:9]:
        print(r['Name'].replace('(anonymous namespace)::', '')[:30].ljust(32), r['Calls'].rjust(4),
              '%10.3f ms total %9.3f ms avg %5.1f%%' % (float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e6, float(r['Percentage'])))
PY
}
{
echo "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --frames-in-flight 1 --steps 8 --warmup 2"
echo "(the culled frames and, after them, the exhaustive frames of the same run: k_mesh belongs to the latter)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/one -- python3 bench.py --no-cpu-baseline --frames-in-flight 1 --steps 8 --warmup 2 > $OUT/one.log 2>&1 || { echo failed; tail -3 $OUT/one.log; exit 1; }
summ $OUT/one
grep -v "^[EW]2026" $OUT/one.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'Mpaths/s', d['value'], 's_per_frame_single', d['s_per_frame_single'])"
echo
echo "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --steps 18 --warmup 3   (three frames in flight: kernels of three streams share the device, their durations stretch)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/three -- python3 bench.py --no-cpu-baseline --steps 18 --warmup 3 > $OUT/three.log 2>&1 || { echo failed; tail -3 $OUT/three.log; exit 1; }
summ $OUT/three
grep -v "^[EW]2026" $OUT/three.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'Mpaths/s', d['value'])"
echo
echo "One culled frame of the first run that was not a sampling step, launch by launch (tools/frame_trace.py):"
python3 tools/frame_trace.py $OUT/one 10
echo
echo "tools/scaling_probe.py 1 2 4 8 (a rank's share of the frame on one GPU, one-rank RCCL group):"
for n in 1 2 4 8; do timeout -k 10 200 python3 tools/scaling_probe.py $n 2>/dev/null | grep "^share"; done
} > $OUT/kernel_summary.txt 2>&1
cp $OUT/one/*/*kernel_stats.csv $OUT/kernel_stats_one_frame_at_a_time.csv
cp $OUT/three/*/*kernel_stats.csv $OUT/kernel_stats_three_frames_in_flight.csv
cat $OUT/kernel_summary.txt
