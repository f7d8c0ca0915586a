"""Same-process A/B: the two encoders as one paired node (batched conv launches) vs two nodes on two streams."""
import sys, os, copy, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-point-clouds-autocomplete_amd"))
import bench
from hyperpocket_amd.core.setup import weights_init
from hyperpocket_amd.model.full_model import FullModel
from hyperpocket_amd.core.engine import TrainEngine
torch.manual_seed(0)
m = FullModel(copy.deepcopy(bench.MODEL_CFG)); m.apply(weights_init); m = m.cuda()
eng = TrainEngine(m, emd_coef=0.05)
ex, mi, gt = bench.synth_batch(64, 1024, torch.device("cuda"), 1)
def run(n):
    for _ in range(n): eng.step(ex, mi, gt, 1)
    eng.finish_pending(); torch.cuda.synchronize()
def t(n=40):
    run(5); t0 = time.perf_counter(); run(n); return (time.perf_counter() - t0) / n * 1e3
for rep in range(4):
    for paired in (True, False):
        m.paired_encoders = paired
        print(f"paired={paired}: step {t():.4f} ms", flush=True)
