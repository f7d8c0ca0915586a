"""GPU box: cProfile of the drop-in FullModel forward alone (Python side), top entries by own time."""
import copy, cProfile, os, pstats, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-point-clouds-autocomplete_amd"))
import bench
from hyperpocket_amd.core.setup import weights_init
from hyperpocket_amd.model.full_model import FullModel
from hyperpocket_amd.optim import FlatAdam
device = torch.device("cuda")
torch.manual_seed(2020)
model = FullModel(copy.deepcopy(bench.MODEL_CFG)); model.apply(weights_init); model = model.to(device)
opt = FlatAdam(model, lr=1e-4)
ex, mi, gt = bench.synth_batch(64, 1024, device, 2020)
model.train()
def fwd():
    e, m = ex.clone(), mi.clone()
    return model(e, m, list(gt.shape), 1, device)
for _ in range(10): fwd()
torch.cuda.synchronize()
N = 200
t0 = time.perf_counter()
for _ in range(N):
    fwd()
    if _ % 8 == 7: torch.cuda.synchronize()
torch.cuda.synchronize()
print(f"forward host+device loop: {(time.perf_counter() - t0) / N * 1e3:.3f} ms per call")
pr = cProfile.Profile(); pr.enable()
for _ in range(N):
    fwd()
    if _ % 8 == 7: torch.cuda.synchronize()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
