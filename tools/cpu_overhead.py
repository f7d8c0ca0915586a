import copy, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-point-clouds-autocomplete_amd"))
import bench
from hyperpocket_amd.core.engine import TrainEngine
from hyperpocket_amd.core.setup import weights_init
from hyperpocket_amd.model.full_model import FullModel
torch.manual_seed(2020)
model = FullModel(copy.deepcopy(bench.MODEL_CFG)); model.apply(weights_init); model = model.cuda()
eng = TrainEngine(model, emd_coef=0.05)
for B in (64, 32):
    ex, mi, gt = bench.synth_batch(B, 1024, torch.device("cuda"), 1)
    for _ in range(5): eng.step(ex, mi, gt, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): eng.step(ex, mi, gt, 1)
    t1 = time.perf_counter()          # CPU enqueue time (GPU may lag behind)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"B={B}: cpu enqueue {1e3*(t1-t0)/20:.2f} ms/step, total {1e3*(t2-t0)/20:.2f} ms/step")
