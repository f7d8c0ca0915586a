#!/usr/bin/env python3
"""Turns a rocprofv3 `--kernel-trace --stats --output-format csv` directory into the markdown
summary committed under profiles/ (per-kernel calls / total / average / share)."""
import csv
import glob
import os
import sys


def main(src, dst, title, steps):
    f = max(glob.glob(os.path.join(src, "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(dst, "w") as o:
        o.write(f"# {title}\n\n")
        o.write(f"Source: `rocprofv3 --kernel-trace --stats --output-format csv` ({os.path.basename(f)}); "
                f"{steps} steps in the trace; total kernel time {tot / 1e6:.2f} ms = {tot / 1e6 / steps:.3f} ms/step.\n\n")
        o.write("| kernel | calls | total ms | avg µs | min µs | max µs | % |\n|---|---:|---:|---:|---:|---:|---:|\n")
        for r in rows:
            if float(r["Percentage"]) < 0.05:
                continue
            name = r["Name"].replace("(anonymous namespace)::", "").replace("|", "\\|")
            if len(name) > 96:
                name = name[:93] + "..."
            o.write(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.1f} | "
                    f"{float(r['MinNs']) / 1e3:.1f} | {float(r['MaxNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |\n")
    print("wrote", dst)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]))
