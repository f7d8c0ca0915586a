#!/usr/bin/env python3
"""VALU-issue model of the EMD sweep family, from the ISA hipcc emits for csrc/emd.hip (runs without a GPU).

    python tools/emd_issue_model.py [--batch 64] [--n 2048] > profiles/r03_emd_issue_model.json

The sweeps are bound by vector-instruction ISSUE (scalar-path candidates, no LDS/HBM pressure): every (row,
candidate-pair) costs a fixed instruction sequence.  For each kernel instance one hp_emd_forward(grad1=NULL, grad2)
call launches, this script finds the software-pipelined main loop in the assembly, counts its vector instructions and
prices them with MI355X_MICROARCH.md's "vector-instruction ISSUE cost" row: transcendentals (v_exp/v_rsq/v_sqrt/v_rcp/
v_log) 8 cycles per wave-instruction, every other VALU op (v_pk_* included) 4.  Wave-instructions per call follow from
the launch geometry (emd.hip: 256-thread workgroups = 4 waves = the 4 candidate ranges of 64*R rows).
bench.py divides the total by the call's measured duration -> `roofline_emd`.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-point-clouds-autocomplete_amd"))
CSRC = os.path.join(ROOT, "3d-point-clouds-autocomplete_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
CXXFILT = "c++filt"
TRANS = ("v_exp_", "v_rsq_", "v_sqrt_", "v_rcp_", "v_log_", "v_sin_", "v_cos_")


def kernels_of(asm):
    """{demangled name: [instruction lines]} for every kernel symbol in the assembly."""
    out, cur = {}, None
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            out[cur] = []
        elif line.startswith(".Lfunc_end"):
            cur = None
        elif cur is not None:
            out[cur].append(line)
    names = subprocess.run([CXXFILT], input="\n".join(out), capture_output=True, text=True, check=True).stdout.split("\n")
    return {n.strip(): body for n, body in zip(names, out.values())}


def main_loop(body):
    """The innermost loop with the most vector instructions: lines between a `.LBBx_y:` label and the backward branch."""
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    best = None
    for i, l in enumerate(body):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            seg = body[labels[m.group(1)]:i]
            nv = sum(1 for s in seg if re.match(r"\s+v_", s))
            if best is None or nv > best[0]:
                best = (nv, seg)
    return best[1]


def price(seg):
    ops = [s.split()[0] for s in seg if re.match(r"\s+v_", s)]
    trans = sum(1 for o in ops if o.startswith(TRANS))
    packed = sum(1 for o in ops if o.startswith("v_pk_"))
    other = len(ops) - trans - packed
    return {"valu_instructions": len(ops), "transcendental": trans, "packed_f32": packed, "other_valu": other,
            "scalar_loads": sum(1 for s in seg if re.match(r"\s+s_load_", s)),
            "issue_cycles_per_iteration": 8 * trans + 4 * (packed + other)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--n", type=int, default=2048)
    args = ap.parse_args()
    import build as hp_build                                                    # same flags as the shipped library
    with tempfile.TemporaryDirectory() as td:
        s = os.path.join(td, "emd.s")
        flags = [f for f in hp_build.FLAGS if f not in ("-fPIC",)]
        subprocess.check_call([HIPCC] + flags + ["-S", "--cuda-device-only", "-o", s, os.path.join(CSRC, "emd.hip")],
                              stderr=subprocess.DEVNULL)
        ks = kernels_of(open(s).read())
    b_call, n = args.batch, args.n
    parts, wg_rows = 4, 64
    # emd.hip emd_forward_impl (round 5): a batch whose halves still fill the chip runs as TWO chains of half the clouds; the
    # instance heuristics see a chain's clouds, the call's wave-instructions are those of both chains
    half = b_call // 2
    chains = 2 if half * -(-n // wg_rows) * 4 >= 2048 else 1
    b = half if chains == 2 else b_call      # (an odd batch's second chain has one cloud more: not modelled)

    def pick(rows, cap):                                                        # emd.hip run_levels(): pick()
        r = cap
        while r > 1:
            if b * -(-rows // (r * wg_rows)) * 4 >= 2048:
                return r
            r >>= 1
        return 1
    r1, r2 = pick(n, 2), pick(n, 4)
    g2 = 2 if b * -(-n // (2 * wg_rows)) * 4 >= 2048 else 1
    pad = -(-n // 64) * 64
    cand = pad // parts                                                         # candidates per wave (its range)
    plan = [  # (kernel instance, launches per call, rows per lane, candidates per loop iteration)
        (f"emd_rows1_kernel<false, true, {r1}>", 1, r1, 16),
        (f"emd_rows1_kernel<true, true, {r1}>", 8, r1, 16),
        # (the last level's phase 3, emd_rows1_kernel<true, false, .>, is not launched on this path: no reader)
        (f"emd_rows2_kernel<{r2}>", 9, r2, 16),
        (f"emd_grad2_kernel<true, {g2}, true>", 1, g2, 4),      # (the final sweep with derived exponentials: round 5)
    ]
    total, rows_out = 0, []
    for name, launches, R, per_iter in plan:
        key = [k for k in ks if re.sub(r"\s+", "", name) in re.sub(r"\s+", "", k)]
        assert len(key) == 1, (name, list(ks))
        p = price(main_loop(ks[key[0]]))
        waves = chains * b * -(-n // (wg_rows * R)) * parts
        iters = cand // per_iter
        cyc = launches * waves * iters * p["issue_cycles_per_iteration"]
        total += cyc
        p.update({"kernel": name, "launches_per_call": launches, "rows_per_lane": R, "waves_per_launch": waves,
                  "loop_iterations_per_wave": iters, "candidates_per_iteration": per_iter,
                  "issue_cycles_per_row_candidate": round(p["issue_cycles_per_iteration"] / (per_iter * R), 3),
                  "issue_cycles_per_call": cyc})
        rows_out.append(p)
    peak = 1024 * 2.4e9
    print(json.dumps({"what": "VALU issue cycles of the main loops of one hp_emd_forward(grad1=NULL, grad2) call; "
                              "transcendental = 8 cycles, any other vector op = 4 (MI355X_MICROARCH.md, issue-cost row)",
                      "batch": b_call, "chains": chains, "clouds_per_chain": b, "n": n, "kernels": rows_out, "issue_cycles_per_call": total,
                      "floor_ms_at_2.4GHz_1024_SIMDs": round(total / peak * 1e3, 4),
                      "hbm_bytes_per_call": None}, indent=1))


if __name__ == "__main__":
    main()
