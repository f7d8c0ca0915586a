"""Share of a reference file's code lines that appear verbatim (whitespace-normalised, comments and blank lines
dropped) in the same-named file of this package — the measure VERDICT r04 used for the model shells.
Runs in the build container only (reads /root/reference)."""
import os
import re
import sys

REF = "/root/reference"
PKG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "3d-point-clouds-autocomplete_amd", "hyperpocket_amd")


def code_lines(path):
    out = []
    for ln in open(path, encoding="utf-8", errors="replace"):
        ln = re.sub(r"\s+", " ", ln.split("#")[0]).strip()
        if ln and ln not in ('"""', "'''"):
            out.append(ln)
    return out


def main():
    rows = []
    for root, _, files in os.walk(PKG):
        for f in files:
            if not f.endswith(".py"):
                continue
            mine = os.path.join(root, f)
            rel = os.path.relpath(mine, PKG)
            ref = os.path.join(REF, rel)
            if not os.path.exists(ref):
                continue
            r, m = code_lines(ref), set(code_lines(mine))
            if not r:
                continue
            hit = sum(1 for ln in r if ln in m)
            rows.append((hit / len(r), hit, len(r), rel))
    for frac, hit, n, rel in sorted(rows, reverse=True):
        print(f"{frac:5.0%}  {hit:3d}/{n:3d}  {rel}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
