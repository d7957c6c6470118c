#!/usr/bin/env python3
"""How busy was the device?  From a rocprofv3 kernel trace: the time at least one kernel was running
(union of the kernel intervals), the sum of the kernel durations (above the union when kernels of several
streams overlap) and the idle time, over the last `frames` frames (a frame starts with k_generate or
k_trace_spheres).

    python tools/trace_busy.py <dir with *_kernel_trace.csv> [frames=30]
"""
import csv
import glob
import os
import sys

f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)[-1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))), key=lambda r: r[0])
starts = [i for i, r in enumerate(rows) if 'k_generate' in r[2] or 'k_trace_spheres' in r[2]]
first = starts[-frames - 1]
last = starts[-1]
window = rows[first:last]
t0, t1 = window[0][0], rows[last][0]
union = 0
cur_s, cur_e = window[0][0], window[0][1]
for s, e, _ in window[1:]:
    if s > cur_e:
        union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += min(cur_e, t1) - cur_s
total = sum(min(e, t1) - s for s, e, _ in window)
wall = t1 - t0
print(f"{frames} frames in {wall / 1e3:.1f} us: {wall / frames / 1e3:.1f} us per frame; a kernel running {union / wall:.1%} of the time, "
      f"kernel durations sum to {total / frames / 1e3:.1f} us per frame ({total / union:.2f} kernels at a time while busy)")
by = {}
for s, e, n in window:
    k = n.replace('(anonymous namespace)::', '').replace('void ', '')[:24]
    by[k] = by.get(k, 0) + (min(e, t1) - s)
for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:8]:
    print(f"  {k:26s}{v / frames / 1e3:9.1f} us per frame")
