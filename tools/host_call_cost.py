"""GPU box: host time of the C entry points of the drop-in forward (ctypes call only, device idle before each block of calls) —
what a hipGraph of their launches could save at most on the reference's own loop."""
import copy, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-point-clouds-autocomplete_amd"))
import bench
from hyperpocket_amd import _lib
from hyperpocket_amd.core.setup import weights_init
from hyperpocket_amd.model.full_model import FullModel
from hyperpocket_amd.optim import FlatAdam
device = torch.device("cuda")
torch.manual_seed(2020)
model = FullModel(copy.deepcopy(bench.MODEL_CFG)); model.apply(weights_init); model = model.to(device)
opt = FlatAdam(model, lr=1e-4)
ex, mi, gt = bench.synth_batch(64, 1024, device, 2020)
model.train()
orig = _lib.call
acc = {}
def timed_call(name, *a):
    t0 = time.perf_counter(); r = orig(name, *a); dt = time.perf_counter() - t0
    s = acc.setdefault(name, [0, 0.0]); s[0] += 1; s[1] += dt
    return r
import hyperpocket_amd.ops as ops
for mod in (ops, _lib):
    if hasattr(mod, "call"): setattr(mod, "call", timed_call)
for it in range(25):
    if it == 5:
        acc.clear()
    torch.cuda.synchronize()
    opt.zero_grad()
    rec, logvar, mu = model(ex.clone(), mi.clone(), list(gt.shape), 1, device)
    loss = rec.sum() * 1e-3 + mu.sum() * 1e-3 + logvar.sum() * 1e-3
    torch.cuda.synchronize()
    loss.backward()
    opt.step()
torch.cuda.synchronize()
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:42s} {n // 20:3d} calls/iter  {t / n * 1e6:7.1f} us per call  {t / 20 * 1e6:8.1f} us per iteration")
