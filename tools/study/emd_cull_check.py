"""GPU dev check of the k-d ordered / culling EMD sweeps (hp_emd_set_cull): agreement with the un-ordered path, validity of the
permutation the order kernel leaves in the workspace, and timings at the bench shape in the regimes of the error map."""
import ctypes, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "3d-point-clouds-autocomplete_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from hyperpocket_amd._lib import call, current_stream, load_library
lib = load_library()
lib.hp_emd_partials_floats.restype = ctypes.c_long
f32 = dict(device="cuda", dtype=torch.float32)


def emd(a, c, want1, want2, keep=None):
    A, C = torch.as_tensor(a).cuda().contiguous(), torch.as_tensor(c).cuda().contiguous()
    b, n, m = A.shape[0], A.shape[1], C.shape[1]
    temp = torch.empty((b, 2 * (n + m)), **f32)
    ws = torch.empty((lib.hp_approxmatch_workspace_floats(b, n, m),), **f32)
    part = torch.empty((lib.hp_emd_partials_floats(b, n, m),), **f32)
    cost = torch.empty((b,), **f32)
    g1 = torch.full((b, n, 3), float("nan"), **f32) if want1 else None
    g2 = torch.full((b, m, 3), float("nan"), **f32) if want2 else None
    call("hp_emd_forward", b, n, m, A, C, temp, ws, part, cost, g1, g2, current_stream(A.device))
    torch.cuda.synchronize()
    if keep is not None:
        keep["ws"] = ws
    return cost, g1, g2


def timeit(a, c, iters=20):
    A, C = torch.as_tensor(a).cuda().contiguous(), torch.as_tensor(c).cuda().contiguous()
    b, n, m = A.shape[0], A.shape[1], C.shape[1]
    temp = torch.empty((b, 2 * (n + m)), **f32)
    ws = torch.empty((lib.hp_approxmatch_workspace_floats(b, n, m),), **f32)
    part = torch.empty((lib.hp_emd_partials_floats(b, n, m),), **f32)
    cost = torch.empty((b,), **f32)
    g2 = torch.empty((b, m, 3), **f32)
    st = current_stream(A.device)
    for _ in range(5):
        call("hp_emd_forward", b, n, m, A, C, temp, ws, part, cost, None, g2, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call("hp_emd_forward", b, n, m, A, C, temp, ws, part, cost, None, g2, st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    r = np.random.RandomState(5)
    print("== agreement cull=K vs cull=0")
    for (b, n, m) in [(3, 96, 96), (2, 200, 330), (5, 330, 200), (4, 1024, 1024), (2, 2048, 2048), (2, 3000, 2048), (66, 2048, 2048)]:
        a = r.rand(b, n, 3).astype(np.float32) - 0.5
        c = r.rand(b, m, 3).astype(np.float32) - 0.5
        lib.hp_emd_set_cull(0)
        c0, g10, g20 = emd(a, c, True, True)
        for K in (1, 4, 9):
            lib.hp_emd_set_cull(K)
            keep = {}
            c1, g11, g21 = emd(a, c, True, True, keep)
            rel = ((c1 - c0).abs() / c0.abs().clamp_min(1e-30)).max().item()
            e1 = (g11 - g10).abs().max().item(); e2 = (g21 - g20).abs().max().item()
            fin = bool(torch.isfinite(g11).all() and torch.isfinite(g21).all())
            # permutation validity
            per = lib.hp_approxmatch_workspace_floats(1, n, m)
            w = keep["ws"].view(b, per)
            NP, MP = (n + 63) // 64 * 64, (m + 63) // 64 * 64
            off = (NP + 8) * 4 + (MP + 8) * 4 + (MP + 8) + (NP + 8) * 16 + (MP + 8) * 16
            pl = w[:, off:off + NP].view(torch.int32)[:, :n].cpu().numpy(); pr = w[:, off + NP:off + NP + MP].view(torch.int32)[:, :m].cpu().numpy()
            okp = all(np.array_equal(np.sort(p), np.arange(n)) for p in pl) and all(np.array_equal(np.sort(p), np.arange(m)) for p in pr)
            print(f"  b={b} n={n} m={m} cull={K}: cost rel {rel:.2e}  g1 max {e1:.2e} g2 max {e2:.2e} finite={fin} perm_ok={okp}")
    from emd_cull_share import regimes
    print("== timing B=64 N=2048")
    for name, (gt, rec) in regimes(per=64).items():
        line = f"  {name:28s}"
        for K in (0, 1, 2, 3, 4, 5):
            lib.hp_emd_set_cull(K)
            line += f"  cull={K}: {timeit(gt, rec):.3f} ms"
        print(line, flush=True)
    lib.hp_emd_set_cull(4)


if __name__ == "__main__":
    main()
