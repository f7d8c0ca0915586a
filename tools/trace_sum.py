import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[2])
tot = {}
for r in rows:
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:40]
    tot.setdefault(k, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
s = 0
for k, v in sorted(tot.items(), key=lambda kv: -sum(kv[1])):
    if 'at::native' in k: continue
    print(f"{k:42s} n/it {len(v)/n:5.1f}  us/it {sum(v)/n:8.1f}  median {sorted(v)[len(v)//2]:7.1f}")
    s += sum(v) / n
print('kernel us per iteration (ex torch elementwise):', round(s, 1))
