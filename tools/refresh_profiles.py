#!/usr/bin/env python3
"""gpurun_out/final (tools/final_measure.sh) -> the round's summaries committed under profiles/ (HP_ROUND, default r04)."""
import csv, glob, json, os, shutil, subprocess, sys
from collections import defaultdict
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(R, "gpurun_out", "final")
P = os.path.join(R, "profiles")
py = sys.executable
RD = os.environ.get("HP_ROUND", "r06")
PRE = int(os.environ.get("HP_PRECONDITION", "400"))      # bench.py --precondition: untimed steps in every step trace
RN = RD.lstrip("r0")


def per_kernel(d):
    """{kernel: ({counter: per-dispatch average}, dispatches, avg_us)} of one --pmc pass."""
    f = max(glob.glob(os.path.join(F, d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
    acc, disp, dur = defaultdict(lambda: defaultdict(float)), defaultdict(set), defaultdict(float)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in disp[k]:
            disp[k].add(r["Dispatch_Id"])
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return {k: ({c: v / len(disp[k]) for c, v in acc[k].items()}, len(disp[k]), dur[k] / len(disp[k])) for k in acc}


def one(d, substr):
    ks = per_kernel(d)
    k = [k for k in ks if substr in k]
    assert len(k) == 1, (d, substr, list(ks))
    return ks[k[0]]


for src, dst in (("bench_n1.json", "bench_n1.json"), ("bench_chamfer.json", "bench_n1_chamfer_only.json"),
                 ("bench_b32.json", "bench_n1_b32_config1.json"),
                 ("bench_stress.json", "bench_chamfer_stress_n8192.json"), ("roofline_events.json", "roofline_hip_events.txt")):
    shutil.copy(os.path.join(F, src), os.path.join(P, f"{RD}_{dst}"))
summ = os.path.join(R, "tools", "summarize_profile.py")
subprocess.check_call([py, summ, os.path.join(F, "step"), os.path.join(P, f"{RD}_step_kernel_stats.md"),
                       f"Round {RN} — full step: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-extras "
                       "--no-cpu-baseline (B=64, Chamfer+EMD; " + str(PRE) + " pre-conditioning + 7 engine steps in the trace)", str(PRE + 7)])
subprocess.check_call([py, summ, os.path.join(F, "roof"), os.path.join(P, f"{RD}_roofline_kernel_stats.md"),
                       f"Round {RN} — roofline launch alone: rocprofv3 --kernel-trace --stats -- python3 bench.py --roofline-minimal "
                       "(encoder conv5: M=65536, N=K=512; 200 warm-up + 100 timed launches, back to back)", "1"])
subprocess.check_call([py, summ, os.path.join(F, "roof_emd"), os.path.join(P, f"{RD}_roofline_emd_kernel_stats.md"),
                       f"Round {RN} — the `roofline` object's command alone: rocprofv3 --kernel-trace --stats -- python3 bench.py --roofline-emd-only "
                       "(hp_emd_forward at B=64, n=m=2048, noisy-copy regime: 10 + 20 calls with the culling sweeps, 10 + 20 with hp_emd_set_cull(0), 1 for the culled share; per shipped call 2 emd_order + 12 culling + 24 plain level launches + 1 emd_grad2 + 1 finish)", "61"])
subprocess.check_call([py, summ, os.path.join(F, "stress"), os.path.join(P, f"{RD}_chamfer_n8192_kernel_stats.md"),
                       f"Round {RN} — BASELINE configs[4] per-GPU shape: rocprofv3 --kernel-trace --stats -- python3 bench.py --workload "
                       "chamfer-stress --steps 5 --warmup 2 --no-extras (B=64, N=8192, Chamfer forward+backward; 7 steps)", "7"])
pm = os.path.join(R, "tools", "pmc_summary.py")
subprocess.check_call([py, pm, os.path.join(P, f"{RD}_pmc_emd_kernels.md"),
                       f"Round {RN} — PMC view of the kernels of hp_emd_forward (bench.py --roofline-emd-only, shipped sweeps: 31 calls at B=64, n=m=2048, noisy-copy regime)",
                       os.path.join(F, "emd_pmc_FETCH_SIZE"), os.path.join(F, "emd_pmc_WRITE_SIZE"), os.path.join(F, "emd_pmc_BUSY")])
subprocess.check_call([py, pm, os.path.join(P, f"{RD}_pmc_chamfer_n8192.md"),
                       f"Round {RN} — PMC view of the Chamfer stress kernels (bench.py --workload chamfer-stress, B=64, N=8192)",
                       os.path.join(F, "stress_pmc_FETCH_SIZE"), os.path.join(F, "stress_pmc_WRITE_SIZE"), os.path.join(F, "stress_pmc_BUSY")])

# ---- dominant GEMM (roofline.traffic)
k = "conv_pp_kernel<1, 2, 2>"
fe, n, _ = one("roof_pmc_FETCH_SIZE", k)
wr, _, _ = one("roof_pmc_WRITE_SIZE", k)
bu, _, us = one("roof_pmc_BUSY", k)
rd = fe["FETCH_SIZE"] * 1024 * 2
out = {
    "kernel": "conv_pp_kernel<1, 2, 2> (both operands as f16 piece pairs, DMA-staged; MFMA 32x32x16 x3; encoder conv5 + fused max-pool, M=65536 N=512 K=512)",
    "command": "rocprofv3 --kernel-trace --pmc <counters> --output-format csv -- python3 bench.py --roofline-minimal   (one pass per "
               "counter group: FETCH_SIZE | WRITE_SIZE | GRBM_GUI_ACTIVE SQ_*; tools/final_measure.sh)",
    "launches_averaged": n, "FETCH_SIZE_KB_raw": round(fe["FETCH_SIZE"], 1),
    "FETCH_SIZE_correction": "x2 (gfx950 counts 128-B requests at 64 B: MI355X_MICROARCH.md §HBM)",
    "WRITE_SIZE_KB": round(wr["WRITE_SIZE"], 1), "hbm_read_bytes_corrected": int(rd),
    "hbm_write_bytes": int(wr["WRITE_SIZE"] * 1024), "hbm_bytes_per_launch": int(rd + wr["WRITE_SIZE"] * 1024),
    "algorithmic_bytes_per_launch": (65536 * 512 + 512 * 512 + 512 + 2 * 512 * 512) * 4,
    "GRBM_GUI_ACTIVE_sum_over_8_xcd": bu["GRBM_GUI_ACTIVE"], "SQ_VALU_MFMA_BUSY_CYCLES": bu["SQ_VALU_MFMA_BUSY_CYCLES"],
    "mfma_pipe_busy_frac": round(bu["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * bu["GRBM_GUI_ACTIVE"] / 8), 4),
    "mfma_pipe_busy_is": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles): the fraction of the f16 matrix pipe's cycles AT THE CLOCK THE CHIP HELD",
    "clock_GHz_in_profiled_pass": round(bu["GRBM_GUI_ACTIVE"] / 8 / (us * 1e-6) / 1e9, 3),
    "note": "profiled passes run at a lower clock than un-profiled ones; compare fractions, not times",
}
json.dump(out, open(os.path.join(P, f"{RD}_pmc_gemm_conv5.json"), "w"), indent=1)

# ---- Chamfer stress kernel (roofline.traffic of --workload chamfer-stress)
k = "nn_distance_kernel<4, true>"
fe, n, _ = one("stress_pmc_FETCH_SIZE", k)
wr, _, _ = one("stress_pmc_WRITE_SIZE", k)
bu, _, us = one("stress_pmc_BUSY", k)
rd = fe["FETCH_SIZE"] * 1024 * 2
sim = 1024 * bu["GRBM_GUI_ACTIVE"] / 8
out = {
    "kernel": "nn_distance_kernel<4,true> (both directed passes of the Chamfer forward), B=64, N=8192", "batch": 64, "n": 8192,
    "launches_averaged": n, "avg_us_profiled": round(us, 1),
    "hbm_read_bytes_corrected": int(rd), "hbm_write_bytes": int(wr["WRITE_SIZE"] * 1024),
    "hbm_bytes_per_launch": int(rd + wr["WRITE_SIZE"] * 1024), "algorithmic_bytes_per_launch": 64 * (2 * 8192 * 12 + 2 * 8192 * 8),
    "hbm_GBps": round((rd + wr["WRITE_SIZE"] * 1024) / (us * 1e-6) / 1e9, 1),
    "SQ_INSTS_VALU": bu["SQ_INSTS_VALU"], "SQ_ACTIVE_INST_VALU": bu["SQ_ACTIVE_INST_VALU"],
    "valu_issue_busy_frac": round(bu["SQ_ACTIVE_INST_VALU"] * 4 / sim, 4),
    "clock_GHz_in_profiled_pass": round(bu["GRBM_GUI_ACTIVE"] / 8 / (us * 1e-6) / 1e9, 3),
    "valu_instructions_per_point_pair": round(bu["SQ_INSTS_VALU"] * 64 / (2.0 * 64 * 8192 * 8192), 3),
    "note": "VALU issue busy = SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE/8): the kernel is bound by vector "
            "issue at the clock the chip holds under this load, not by HBM",
}
json.dump(out, open(os.path.join(P, f"{RD}_pmc_chamfer_n8192.json"), "w"), indent=1)

# ---- EMD family: HBM bytes per hp_emd_forward call + measured VALU issue cycles against the model
fes, wrs, bus = per_kernel("emd_pmc_FETCH_SIZE"), per_kernel("emd_pmc_WRITE_SIZE"), per_kernel("emd_pmc_BUSY")
steps = 31      # hp_emd_forward calls in `bench.py --roofline-emd-only` with HP_BENCH_EMD_NO_UNCULL=1
tot_bytes = tot_valu = tot_us = tot_gui = 0.0
rows = []
for kname in sorted(k for k in bus if k.startswith("void emd_") or k.startswith("emd_")):
    c, n, us = bus[kname]
    per_step = n / steps
    b = (fes[kname][0]["FETCH_SIZE"] * 2048 + wrs[kname][0]["WRITE_SIZE"] * 1024) * per_step
    tot_bytes += b
    tot_valu += c["SQ_ACTIVE_INST_VALU"] * 4 * per_step
    tot_us += us * per_step
    tot_gui += c["GRBM_GUI_ACTIVE"] / 8 * per_step
    rows.append({"kernel": kname, "launches_per_call": per_step, "avg_us_profiled": round(us, 1),
                 "valu_issue_busy_frac": round(c["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * c["GRBM_GUI_ACTIVE"] / 8), 4),
                 "hbm_MB_per_launch": round((fes[kname][0]["FETCH_SIZE"] * 2048 + wrs[kname][0]["WRITE_SIZE"] * 1024) / 1e6, 2)})
mp = os.path.join(P, f"{RD}_emd_issue_model.json")
if not os.path.exists(mp):      # the static instruction-stream model of the un-culled sweeps: carried over from the round that made it
    prev = sorted(glob.glob(os.path.join(P, "r[0-9][0-9]_emd_issue_model.json")))[-1]
    shutil.copy(prev, mp)
model = json.load(open(mp))
model["hbm_bytes_per_call"] = int(tot_bytes)
model["pmc"] = {"command": "rocprofv3 --kernel-trace --pmc ... -- python3 bench.py --roofline-emd-only (HP_BENCH_EMD_NO_UNCULL=1; "
                           "tools/final_measure.sh): the shipped call (Hilbert order + culling sweeps), noisy-copy regime, 31 calls",
                "SQ_ACTIVE_INST_VALU_x4_cycles_per_call": int(tot_valu),
                "model_over_measured_valu_cycles": round(model["issue_cycles_per_call"] / tot_valu, 4),
                "valu_issue_busy_frac_of_kernel_time": round(tot_valu / (1024 * tot_gui), 4),
                "kernel_time_us_per_call_profiled": round(tot_us, 1), "kernels": rows}
json.dump(model, open(mp, "w"), indent=1)
print(json.dumps(model["pmc"], indent=1)[:1500])
