"""Summarise a rocprofv3 --kernel-trace CSV: calls / average / minimum duration per (kernel, grid size).
usage: python tools/trace_summary.py <dir or *_kernel_trace.csv> [name filter]"""
import csv, glob, os, sys
from collections import defaultdict

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
files = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, '**', '*kernel_trace.csv'), recursive=True)
acc = defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name']
        if flt and flt not in name:
            continue
        short = name.split('(')[0].replace('void ', '').replace('dpk::', '')
        grid = int(r.get('Grid_Size_X', r.get('Grid_Size', 0))) // max(1, int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 1))))
        acc[(short[:60], grid, r.get('Grid_Size_Y', '1'))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print('%-60s %8s %4s %7s %10s %10s %10s' % ('kernel', 'blocks', 'gy', 'calls', 'avg_us', 'min_us', 'med_us'))
for (k, g, gy), v in sorted(acc.items(), key=lambda kv: (kv[0][0], kv[0][1])):
    v2 = sorted(v)
    print('%-60s %8d %4s %7d %10.2f %10.2f %10.2f' % (k, g, gy, len(v), sum(v) / len(v), v2[0], v2[len(v2) // 2]))
