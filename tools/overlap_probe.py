#!/usr/bin/env python3
"""Do the VALU-bound EMD sweeps and the MFMA-bound encoder GEMMs overlap when issued on two streams?
T(emd alone), T(encoder forward alone), T(both, two streams) at half the step's batch (B=32)."""
import copy, ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-point-clouds-autocomplete_amd")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from bench import MODEL_CFG
from hyperpocket_amd._lib import call, current_stream, load_library
from hyperpocket_amd.core.setup import weights_init
from hyperpocket_amd.model.full_model import FullModel

torch.cuda.set_device(0)
dev = torch.device("cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = 2048
torch.manual_seed(0)
model = FullModel(copy.deepcopy(MODEL_CFG)); model.apply(weights_init); model = model.to(dev)
gt = torch.rand(B, N, 3, device=dev) - 0.5
rec = (torch.rand(B, N, 3, device=dev) - 0.5) * 1.2
x = (torch.rand(B, N // 2, 3, device=dev) - 0.5).transpose(1, 2)     # channels-first view, as FullModel hands it over
lib = load_library()
lib.hp_emd_partials_floats.restype = ctypes.c_long
f32 = dict(dtype=torch.float32, device=dev)
temp = torch.empty((B, 4 * N), **f32)
ws = torch.empty((max(1, lib.hp_approxmatch_workspace_floats(B, N, N)),), **f32)
epart = torch.empty((max(1, lib.hp_emd_partials_floats(B, N, N)),), **f32)
cost = torch.empty((B,), **f32); g = torch.empty_like(rec)

def emd():
    call("hp_emd_forward", B, N, N, gt, rec, temp, ws, epart, cost, None, g, current_stream(dev))

def enc(reps=2):
    with torch.no_grad():
        for _ in range(reps):
            model.real_encoder(x)
            model.real_encoder(x)

s1, s2 = torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=-1)

def timed(fn, iters=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3

def both():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        emd()
    with torch.cuda.stream(s2):
        enc()
    cur.wait_stream(s1); cur.wait_stream(s2)

te, tg, tb = timed(emd), timed(enc), timed(both)
print(f"B={B}: emd {te:.3f} ms, encoder fwd x4 {tg:.3f} ms, sum {te+tg:.3f}, concurrent {tb:.3f} ms (max {max(te,tg):.3f})")
