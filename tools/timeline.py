#!/usr/bin/env python3
"""Prints the per-launch timeline of the last of N identical iterations in a rocprofv3 kernel trace."""
import csv, glob, sys
d, iters = sys.argv[1], int(sys.argv[2])
f = glob.glob(d + '/*/*_kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
n = len(rows) // iters
last = rows[-n:]
tot = 0
t0 = int(last[0]['Start_Timestamp'])
for r in last:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3; tot += d
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:58]
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:8.1f} {name:58s} grid {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):5d} x{int(r['Grid_Size_Y']):4d} x{int(r['Grid_Size_Z']):3d} {d:7.1f} us")
print('kernel sum', round(tot, 1), 'span', (int(last[-1]['End_Timestamp']) - t0) / 1e3)
