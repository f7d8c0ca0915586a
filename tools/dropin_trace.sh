#!/bin/bash
# GPU box: kernel trace of the reference's own loop over the drop-in modules (tools/dropin_profile.py) -> where the GPU idles
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R=$GRAFT_REPO_ROOT; W=${1:-flat}; O="$R/gpurun_out/dropin_trace_$W"; rm -rf "$O"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/dropin_profile.py $W > $O/log.txt 2>&1
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
python3 - <<PY
import csv, glob
f = glob.glob("$O/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "conv_split_prep" in r["Kernel_Name"]]
a, b = idx[30], idx[40]           # ten steady-state iterations
t0 = int(rows[a]["Start_Timestamp"]); span = (int(rows[b]["Start_Timestamp"]) - t0) / 10e3
end = t0; gaps = {}; busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > end:
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:40]
        gaps[n] = gaps.get(n, 0) + (s - end)
    if e > end:
        busy += e - max(s, end); end = e
print(f"[$W] iteration {span:.1f} us, GPU busy (union) {busy / 10e3:.1f} us, idle {span - busy / 10e3:.1f} us; idle time by the kernel that ends it (us per iteration):")
for n, v in sorted(gaps.items(), key=lambda kv: -kv[1])[:14]:
    print(f"   {v / 10e3:7.1f}  before {n}")
PY
