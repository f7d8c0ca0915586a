#!/usr/bin/env python3
"""Timeline of the last engine step in a rocprofv3 kernel trace of bench.py: per-queue lanes, gaps, overlap."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# step boundaries: adam_kernel launches (several buckets per step); find the last 'sample_points' -> next 'sample_points'
idx = [i for i, r in enumerate(rows) if 'sample_points' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
step = rows[a:b]
t0 = int(step[0]['Start_Timestamp'])
busy = 0
end = t0
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:50]
    gap = (s - end) / 1e3
    print(f"{(s-t0)/1e3:8.1f} q{r['Queue_Id']:>2s} {name:50s} g{int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):5d}x{int(r['Grid_Size_Y']):<4d} {(e-s)/1e3:7.1f}" + (f"   <-- idle {gap:.1f}" if gap > 3 else ""))
    if e > end:
        busy += (e - max(s, end)); end = e
print('span', (end - t0) / 1e3, 'busy(union)', busy / 1e3, 'sum', sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step) / 1e3)
