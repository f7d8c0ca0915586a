"""Per-launch timeline of the last hp_emd_forward call in a rocprofv3 kernel trace (csv): one line per stream."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'emd_init' in n or 'emd_order' in n]
seq = rows[idx[-2]:]
by = {}
for r in seq:
    by.setdefault(r.get('Stream_Id', r.get('Queue_Id')), []).append(r)
t0 = int(seq[0]['Start_Timestamp'])
for q, rs in by.items():
    out = []
    for r in rs:
        n = r['Kernel_Name']
        tag = 'ord' if 'order' in n else 'init' if 'init' in n else 'r1c' if 'rows1_cull' in n else 'r2c' if 'rows2_cull' in n else 'r1' if 'rows1' in n else 'r2' if 'rows2' in n else 'g2' if 'grad2' in n else 'fin'
        out.append(f"{tag}@{(int(r['Start_Timestamp'])-t0)/1000:.0f}+{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000:.0f}")
    print('stream', q, ' '.join(out))
