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
from hyperpocket_amd import ops
flags = [a.split("=") for a in sys.argv[1:]]          # e.g. DEDUP_CRITICAL_ROWS=0  -> also run with that ops flag flipped
for rep in range(3):
    for emd in (0.05, 0.0):
        eng.emd_coef = emd
        print(f"emd={emd}: {t():.4f} ms", flush=True)
        for name, val in flags:
            old = getattr(ops, name)
            setattr(ops, name, type(old)(int(val)))
            print(f"emd={emd} {name}={val}: {t():.4f} ms", flush=True)
            setattr(ops, name, old)
# CPU enqueue time
eng.emd_coef = 0.05
run(5); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): eng.step(ex, mi, gt, 1)
cpu = (time.perf_counter() - t0) / 20 * 1e3
torch.cuda.synchronize()
print(f"cpu enqueue {cpu:.3f} ms/step")
