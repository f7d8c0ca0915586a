import sys, os, copy, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-point-clouds-autocomplete_amd"))
import bench
from hyperpocket_amd import _lib
from hyperpocket_amd.core.setup import weights_init
from hyperpocket_amd.model.full_model import FullModel
L = _lib.load_library()
torch.manual_seed(0)
m = FullModel(copy.deepcopy(bench.MODEL_CFG)); m.apply(weights_init); m = m.cuda()
hn = m.hyper_network
lat = torch.randn(64, 256, device="cuda", requires_grad=True)
w = torch.randn(64, 19011, device="cuda")
n = 50
for _ in range(5):
    th = hn(lat); th.backward(w)
torch.cuda.synchronize()
e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
tf = tb = 0.0
for _ in range(n):
    e0.record(); th = hn(lat); e1.record(); th.backward(w); e2.record()
    torch.cuda.synchronize()
    tf += e0.elapsed_time(e1); tb += e1.elapsed_time(e2)
print(os.environ.get("HP_SKINNY"), os.environ.get("HP_SK_DEBUG"), os.environ.get("HP_SK_SF"), os.environ.get("HP_SK_SX"), f"hypernet fwd {tf / n * 1e3:.1f} us  bwd {tb / n * 1e3:.1f} us", flush=True)
