"""A/B of the persistent skinny-M layer programs (csrc/skinny.hip) against the tiled GEMM launches, same process:
engine step time and the hypernetwork forward+backward alone (HIP events)."""
import sys, os, copy, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-point-clouds-autocomplete_amd"))
import bench
from hyperpocket_amd import _lib
from hyperpocket_amd.core.setup import weights_init
from hyperpocket_amd.model.full_model import FullModel
from hyperpocket_amd.core.engine import TrainEngine
L = _lib.load_library()
torch.manual_seed(0)
m = FullModel(copy.deepcopy(bench.MODEL_CFG)); m.apply(weights_init); m = m.cuda()
eng = TrainEngine(m, emd_coef=0.05)
ex, mi, gt = bench.synth_batch(64, 1024, torch.device("cuda"), 1)
def run(n):
    for _ in range(n): eng.step(ex, mi, gt, 1)
    eng.finish_pending(); torch.cuda.synchronize()
def t(n=40):
    run(5); t0 = time.perf_counter(); run(n); return (time.perf_counter() - t0) / n * 1e3
hn = m.hyper_network
lat = torch.randn(64, 256, device="cuda", requires_grad=True)
w = torch.randn(64, 19011, device="cuda")
def hyper(n=50):
    for _ in range(5):
        th = hn(lat); th.backward(w)
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    tf = tb = 0.0
    for _ in range(n):
        e0.record(); th = hn(lat); e1.record(); th.backward(w); e2.record()
        torch.cuda.synchronize()
        tf += e0.elapsed_time(e1); tb += e1.elapsed_time(e2)
    return tf / n * 1e3, tb / n * 1e3
for rep in range(3):
    for on in (1, 0):
        L.hp_skinny_set_enabled(on)
        f, b = hyper()
        print(f"skinny={on}: step {t():.4f} ms   hypernet fwd {f:.1f} us  bwd {b:.1f} us", flush=True)
