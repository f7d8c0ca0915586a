"""cProfile of the host side of TrainEngine.step (where does the CPU enqueue time go)."""
import copy, cProfile, os, pstats, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-point-clouds-autocomplete_amd"))
import bench
from hyperpocket_amd.core.engine import TrainEngine
from hyperpocket_amd.core.setup import weights_init
from hyperpocket_amd.model.full_model import FullModel
torch.manual_seed(2020)
model = FullModel(copy.deepcopy(bench.MODEL_CFG)); model.apply(weights_init); model = model.cuda()
eng = TrainEngine(model, emd_coef=0.05)
ex, mi, gt = bench.synth_batch(8, 1024, torch.device("cuda"), 1)
for _ in range(10): eng.step(ex, mi, gt, 1)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(50): eng.step(ex, mi, gt, 1)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(25)
