#!/bin/bash
# GPU box: counters of conv_pp_kernel on the conv5 / conv4 shapes (tools/bench_pp.py ONLY=1)
: "${GRAFT_REPO_ROOT:?run this on the GPU box through gpurun}"
R=$GRAFT_REPO_ROOT; O="$R/gpurun_out/pmc_pp"; rm -rf "$O"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp; export ONLY=1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/a -- python3 $R/tools/bench_pp.py > $O/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/b -- python3 $R/tools/bench_pp.py > $O/b.log 2>&1
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
python3 - <<PY
import csv, glob, collections
for tag in "ab":
    f = glob.glob("$O/%s/**/*counter_collection.csv" % tag, recursive=True)
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        if "conv_pp" not in r["Kernel_Name"]: continue
        k = (r["Kernel_Name"][:40], r["Dispatch_Id"])
        d = rows.setdefault(k, {})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
        d["us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    seen = collections.Counter()
    for (k, d), c in rows.items():
        seen[k] += 1
        if seen[k] in (120, 121):
            print(tag, k, " ".join("%s=%.4g" % kv for kv in c.items()))
PY
