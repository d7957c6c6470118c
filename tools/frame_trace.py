"""Per-launch kernel durations of one culled frame from a rocprofv3 kernel trace:
python tools/frame_trace.py <dir with *_kernel_trace.csv> [frame index from the end] [sum]
("sum": one line per kernel instead of one per launch)"""
import csv, glob, sys

f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=lambda p: __import__('os').path.getmtime(p))[-1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [(r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:34], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
starts = [i for i, n in enumerate(names) if n[0].startswith('k_generate') and i + 1 < len(names) and names[i + 1][0].startswith(('k_bin', 'k_group', 'k_mesh', 'k_mark'))]
s = starts[-back]
prev = None
total = 0.0
per_kernel = {}
for n, st, en in names[s:]:
    if prev is not None and n.startswith('k_generate'):
        break
    if 'sum' in sys.argv[3:]:
        key = n.split('(')[0]
        per_kernel.setdefault(key, []).append((en - st) / 1e3)
    else:
        print(f"{n:36s}{(en - st) / 1e3:9.1f} us   gap {((st - prev) / 1e3 if prev else 0):6.1f}")
    total += (en - st) / 1e3
    prev = en
for key, v in per_kernel.items():
    print(f"{key:36s}{sum(v):9.1f} us  x{len(v)}  ({' '.join('%.0f' % x for x in v)})")
print(f"kernel time {total:.1f} us, first start to last end {(prev - names[s][1]) / 1e3:.1f} us")
