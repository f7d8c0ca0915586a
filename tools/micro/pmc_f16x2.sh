#!/bin/bash
# GPU box: counters of the split-GEMM harness (clock under load, MFMA busy, LDS bank conflicts)
: "${GRAFT_REPO_ROOT:?run this on the GPU box through gpurun (GRAFT_REPO_ROOT unset)}"
R=$GRAFT_REPO_ROOT; O="$R/gpurun_out/pmc_f16x2"; rm -rf "$O"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
# the profiled binary is built here, from the source next to this script (it is git-ignored: a committed copy goes stale)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o "$R/tools/micro/gemm_f16x2" "$R/tools/micro/gemm_f16x2.hip" || exit 1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/a -- $R/tools/micro/gemm_f16x2 > $O/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/b -- $R/tools/micro/gemm_f16x2 > $O/b.log 2>&1
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
python3 - <<PY
import csv, glob, collections
for tag in "ab":
    f = glob.glob("$O/%s/**/*counter_collection.csv" % tag, recursive=True)
    kt = glob.glob("$O/%s/**/*kernel_trace.csv" % tag, recursive=True)
    dur = {}
    for r in csv.DictReader(open(kt[0])):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        rows.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    for d, c in list(rows.items())[:40]:
        print(tag, d, "%.1f us" % dur.get(d, 0), " ".join("%s=%.4g" % kv for kv in c.items()))
PY
