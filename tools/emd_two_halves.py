#!/usr/bin/env python3
"""Experiment (GPU box): one hp_emd_forward call on 64 clouds against two calls on 32 clouds each, issued on two streams —
the second chain's launches fill the ramps / tails of the first's (no persistent kernel, no hand-off flags)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
from hyperpocket_amd._lib import call, load_library  # noqa: E402

torch.cuda.set_device(0)
lib = load_library()
lib.hp_emd_partials_floats.restype = ctypes.c_long
B, n = 64, 2048
f32 = dict(dtype=torch.float32, device="cuda")
g = torch.Generator(device="cuda").manual_seed(7)
a = torch.rand(B, n, 3, generator=g, **f32) - 0.5
c = torch.rand(B, n, 3, generator=g, **f32) - 0.5


def bufs(b):
    return (torch.empty((b, 4 * n), **f32), torch.empty((lib.hp_approxmatch_workspace_floats(b, n, n),), **f32),
            torch.empty((lib.hp_emd_partials_floats(b, n, n),), **f32), torch.empty((b,), **f32), torch.empty((b, n, 3), **f32))


full = bufs(B)
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
hb = B // parts
halves = [bufs(hb) for _ in range(parts)]
side = [torch.cuda.Stream(priority=-1) for _ in range(parts - 1)]
cur = torch.cuda.current_stream()


def one():
    t, ws, p, cost, g2 = full
    call("hp_emd_forward", B, n, n, a, c, t, ws, p, cost, None, g2, ctypes.c_void_p(cur.cuda_stream))


def split():
    for s in side:
        s.wait_stream(cur)
    for i in range(parts):
        st = cur if i == 0 else side[i - 1]
        t, ws, p, cost, g2 = halves[i]
        call("hp_emd_forward", hb, n, n, a[i * hb:(i + 1) * hb], c[i * hb:(i + 1) * hb], t, ws, p, cost, None, g2, ctypes.c_void_p(st.cuda_stream))
    for s in side:
        cur.wait_stream(s)


for rows in ((0, 0, 0), (2, 4, 2), (2, 2, 2), (1, 2, 1)):
    lib.hp_emd_set_rows_per_lane(*rows)
    t1 = min(bench.event_time_ms(one, iters=20, warm=10) for _ in range(3))
    t2 = min(bench.event_time_ms(split, iters=20, warm=10) for _ in range(3))
    print(f"rows-per-lane {rows}: one call {t1:.4f} ms | {parts} x {hb} clouds on {parts} streams {t2:.4f} ms", flush=True)
lib.hp_emd_set_rows_per_lane(0, 0, 0)
one(); split(); torch.cuda.synchronize()
print("cost agreement:", torch.equal(full[3], torch.cat([h[3] for h in halves])), "grad:", torch.equal(full[4], torch.cat([h[4] for h in halves])))
