"""GPU box: where an iteration of the reference's own loop (core/epoch_loops.py:15-39 over the drop-in modules) spends its HOST
time — per segment (host clock, no extra syncs) and as a cProfile."""
import copy, cProfile, os, pstats, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-point-clouds-autocomplete_amd"))
import bench
from hyperpocket_amd.core.setup import weights_init
from hyperpocket_amd.losses.champfer_loss import ChamferLoss
from hyperpocket_amd.model.full_model import FullModel
from hyperpocket_amd.optim import FlatAdam
device = torch.device("cuda")
which = sys.argv[1] if len(sys.argv) > 1 else "flat"
torch.manual_seed(2020)
model = FullModel(copy.deepcopy(bench.MODEL_CFG)); model.apply(weights_init); model = model.to(device)
loss_fn = ChamferLoss().to(device)
opt = torch.optim.Adam(model.parameters(), lr=1e-4) if which == "torch" else FlatAdam(model, lr=1e-4)
ex, mi, gt = (t.cpu().pin_memory() for t in bench.synth_batch(64, 1024, device, 2020))
model.train()
seg = {}
def mark(name, t0):
    t = time.perf_counter(); seg[name] = seg.get(name, 0.0) + (t - t0); return t
def iteration():
    t = time.perf_counter()
    opt.zero_grad(); t = mark("zero_grad", t)
    e, m, g = ex.to(device), mi.to(device), gt.to(device); t = mark("h2d", t)
    rec, logvar, mu = model(e, m, list(g.shape), 1, device); t = mark("forward", t)
    loss_r = torch.mean(0.05 * loss_fn(g, rec.permute(0, 2, 1))); t = mark("chamfer", t)
    kld = torch.div(0.5 * (torch.exp(logvar) + torch.square(mu) - 1 - logvar).sum(), e.shape[0])
    total = loss_r + kld; t = mark("kld", t)
    a = kld.item(); t = mark("item1(wait)", t)
    b = loss_r.item(); c = total.item(); t = mark("item2,3", t)
    total.backward(); t = mark("backward", t)
    opt.step(); t = mark("opt.step", t)
for _ in range(10): iteration()
torch.cuda.synchronize(); seg.clear()
N = 50
t0 = time.perf_counter()
for _ in range(N): iteration()
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / N * 1e3
print(f"[{which}] {tot:.3f} ms/iteration; host segments (ms): " + ", ".join(f"{k} {v / N * 1e3:.3f}" for k, v in seg.items()))
pr = cProfile.Profile(); pr.enable()
for _ in range(N): iteration()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
