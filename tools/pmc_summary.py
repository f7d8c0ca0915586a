#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc passes (one directory per pass) -> markdown table / JSON.
usage: pmc_summary.py OUT.md title dir_fetch dir_write dir_busy"""
import csv, glob, json, os, sys
from collections import defaultdict


def load(d):
    f = max(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    dur = defaultdict(float)
    seen = defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
        did = r["Dispatch_Id"]
        if did not in seen[k] and "Start_Timestamp" in r:
            seen[k].add(did)
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return acc, cnt, dur, {k: len(v) for k, v in seen.items()}


def main(out, title, d_fetch, d_write, d_busy):
    fa, fc, fd, fn = load(d_fetch)
    wa, wc, _, _ = load(d_write)
    ba, bc, _, _ = load(d_busy)
    rows = []
    for k in fa:
        n = max(1, fn.get(k, 1))
        fetch_kb = fa[k].get("FETCH_SIZE", 0.0) / n
        write_kb = wa.get(k, {}).get("WRITE_SIZE", 0.0) / max(1, wc.get(k, {}).get("WRITE_SIZE", 1) or 1) * (wc.get(k, {}).get("WRITE_SIZE", 1) / max(1, fn.get(k, 1)))
        gui = ba.get(k, {}).get("GRBM_GUI_ACTIVE", 0.0) / n
        mfma = ba.get(k, {}).get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / n
        valu = ba.get(k, {}).get("SQ_ACTIVE_INST_VALU", 0.0) / n
        us = fd[k] / n
        rd, wr = fetch_kb * 1024 * 2, write_kb * 1024
        rows.append(dict(kernel=k, launches=n, avg_us=us, read_mb=rd / 1e6, write_mb=wr / 1e6,
                         gbps=(rd + wr) / (us * 1e-6) / 1e9 if us else 0.0,
                         valu_busy=valu * 4 / (1024 * gui / 8) if gui else 0.0,
                         mfma_busy=mfma / (1024 * gui / 8) if gui else 0.0, total_us=fd[k]))
    rows.sort(key=lambda r: -r["total_us"])
    with open(out, "w") as o:
        o.write(f"# {title}\n\n")
        o.write("Three separate `rocprofv3 --kernel-trace --pmc` passes (FETCH_SIZE | WRITE_SIZE | GRBM_GUI_ACTIVE + SQ_*), averaged "
                "per kernel name.\nHBM bytes = FETCH_SIZE x 1024 x 2 (gfx950 correction for wide streaming reads — an upper bound "
                "for narrow/strided ones) + WRITE_SIZE x 1024; GB/s = bytes / profiled duration; VALU busy = SQ_ACTIVE_INST_VALU x 4 / "
                "(1024 SIMDs x GRBM_GUI_ACTIVE/8); MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x GRBM_GUI_ACTIVE/8).  Profiled "
                "passes run slower than un-profiled ones.\n\n")
        o.write("| kernel | launches | avg µs | HBM read MB | HBM write MB | HBM GB/s | VALU busy | MFMA busy |\n|---|---:|---:|---:|---:|---:|---:|---:|\n")
        for r in rows:
            if r["total_us"] < 20:
                continue
            o.write(f"| `{r['kernel'][:70].replace('|', '/')}` | {r['launches']} | {r['avg_us']:.1f} | {r['read_mb']:.1f} | {r['write_mb']:.1f} | "
                    f"{r['gbps']:.0f} | {r['valu_busy']:.2f} | {r['mfma_busy']:.2f} |\n")
    json.dump(rows, open(out.replace(".md", ".json"), "w"), indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main(*sys.argv[1:6])
