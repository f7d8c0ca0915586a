#!/usr/bin/env python3
"""gpurun_out/final (tools/final_measure.sh) -> the summaries committed under profiles/."""
import csv, glob, json, os, shutil, subprocess, sys
from collections import defaultdict
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(R, "gpurun_out", "final")
P = os.path.join(R, "profiles")
py = sys.executable


def counters(d, kernel_substr):
    f = glob.glob(os.path.join(F, d, "**", "*_counter_collection.csv"), recursive=True)[0]
    acc, disp = defaultdict(float), set()
    for r in csv.DictReader(open(f)):
        if kernel_substr in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            disp.add(r["Dispatch_Id"])
    return {k: v / len(disp) for k, v in acc.items()}, len(disp)


shutil.copy(os.path.join(F, "bench_n1.json"), os.path.join(P, "r01_bench_n1.json"))
shutil.copy(os.path.join(F, "bench_chamfer.json"), os.path.join(P, "r01_bench_n1_chamfer_only.json"))
shutil.copy(os.path.join(F, "roofline_events.json"), os.path.join(P, "r01_roofline_hip_events.txt"))
subprocess.check_call([py, os.path.join(R, "tools", "summarize_profile.py"), os.path.join(F, "step"),
                       os.path.join(P, "r01_step_kernel_stats_final.md"),
                       "Round 1 — full step: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-extras "
                       "--no-cpu-baseline (B=64, Chamfer+EMD; 7 engine steps, nothing else in the trace)", "7"])
subprocess.check_call([py, os.path.join(R, "tools", "summarize_profile.py"), os.path.join(F, "roof"),
                       os.path.join(P, "r01_roofline_kernel_stats.md"),
                       "Round 1 — roofline launch alone: rocprofv3 --kernel-trace --stats -- python3 bench.py --roofline-only "
                       "(encoder conv5: M=65536, N=K=512; 200 warm-up + 100 timed launches, back to back)", "1"])
subprocess.check_call([py, os.path.join(R, "tools", "pmc_summary.py"), os.path.join(P, "r01_pmc_step_kernels.md"),
                       "Round 1 — PMC view of every kernel of the step (bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline)",
                       os.path.join(F, "step_pmc_FETCH_SIZE"), os.path.join(F, "step_pmc_WRITE_SIZE"), os.path.join(F, "step_pmc_BUSY")])
k = "gemm_kernel<128, 128, 4, 2, 16, 4>"
fe, n = counters("roof_pmc_FETCH_SIZE", k)
wr, _ = counters("roof_pmc_WRITE_SIZE", k)
bu, _ = counters("roof_pmc_BUSY", k)
rd = fe["FETCH_SIZE"] * 1024 * 2
out = {
    "kernel": "gemm_kernel<128,128,4,2,16,4> (fp32 MFMA 32x32x2, encoder conv5 shape M=65536 N=512 K=512, plain C store)",
    "command": "rocprofv3 --kernel-trace --pmc <counters> --output-format csv -- python3 bench.py --roofline-only   (one pass per "
               "counter group: FETCH_SIZE | WRITE_SIZE | GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU; tools/final_measure.sh)",
    "launches_averaged": n,
    "FETCH_SIZE_KB_raw": round(fe["FETCH_SIZE"], 1),
    "FETCH_SIZE_correction": "x2 (gfx950 counts 128-B requests at 64 B: MI355X_MICROARCH.md §HBM)",
    "WRITE_SIZE_KB": round(wr["WRITE_SIZE"], 1),
    "hbm_read_bytes_corrected": int(rd),
    "hbm_write_bytes": int(wr["WRITE_SIZE"] * 1024),
    "hbm_bytes_per_launch": int(rd + wr["WRITE_SIZE"] * 1024),
    "algorithmic_bytes_per_launch": (2 * 65536 * 512 + 512 * 512 + 512) * 4,
    "GRBM_GUI_ACTIVE_sum_over_8_xcd": bu["GRBM_GUI_ACTIVE"],
    "SQ_VALU_MFMA_BUSY_CYCLES": bu["SQ_VALU_MFMA_BUSY_CYCLES"],
    "mfma_pipe_busy_frac": round(bu["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * bu["GRBM_GUI_ACTIVE"] / 8), 4),
    "SQ_ACTIVE_INST_VALU": bu["SQ_ACTIVE_INST_VALU"],
    "note": "profiled passes run at a lower clock than un-profiled ones; compare fractions, not times",
}
json.dump(out, open(os.path.join(P, "r01_pmc_gemm_conv5.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
