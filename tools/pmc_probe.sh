#!/bin/bash
# usage (on the GPU box): tools/pmc_probe.sh OUTDIR "COUNTERS..." -- python3 bench.py ...
# one rocprofv3 --pmc pass with --kernel-trace; prints per-kernel averages of each counter.
R=$GRAFT_REPO_ROOT; O=$1; shift; C=$1; shift; shift
mkdir -p $R/gpurun_out/$O; cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/$O -- "$@" > $R/gpurun_out/$O/run.log 2>&1
find $R/gpurun_out/$O -name "*.db" -delete; find $R/gpurun_out/$O -name "*_agent_info.csv" -delete
python3 - $R/gpurun_out/$O <<'PY'
import csv, glob, os, sys
from collections import defaultdict
f = glob.glob(os.path.join(sys.argv[1], "**", "*_counter_collection.csv"), recursive=True)[0]
acc, n, dur = defaultdict(lambda: defaultdict(float)), defaultdict(set), defaultdict(float)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in n[k]:
        n[k].add(r["Dispatch_Id"]); dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k in sorted(acc, key=lambda k: -dur[k])[:12]:
    c = len(n[k])
    print(f"{k:60s} n={c:4d} avg_us={dur[k]/c:9.1f} " + " ".join(f"{name}={v/c:.4g}" for name, v in sorted(acc[k].items())))
PY
