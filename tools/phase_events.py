"""Un-profiled phase times of one engine step by HIP events on the compute stream (forward | losses | backward | adam)."""
import copy, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-point-clouds-autocomplete_amd"))
import bench
from hyperpocket_amd.core import engine as E
from hyperpocket_amd.core.setup import weights_init
from hyperpocket_amd.model.full_model import FullModel
torch.manual_seed(2020)
model = FullModel(copy.deepcopy(bench.MODEL_CFG)); model.apply(weights_init); model = model.cuda()
eng = E.TrainEngine(model, emd_coef=0.05)
marks = []
def ev(tag):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((tag, e))
orig_losses = eng._losses_and_gradients
def losses(*a, **k):
    ev("fwd_done"); r = orig_losses(*a, **k); ev("loss_done"); return r
eng._losses_and_gradients = losses
orig_bw = torch.autograd.backward
def bw(*a, **k):
    r = orig_bw(*a, **k); ev("bwd_done"); return r
torch.autograd.backward = bw
ex, mi, gt = bench.synth_batch(64, 1024, torch.device("cuda"), 1)
for _ in range(10): eng.step(ex, mi, gt, 1)
torch.cuda.synchronize()
acc = {}
N = 30
for _ in range(N):
    marks.clear(); ev("start"); eng.step(ex, mi, gt, 1); ev("end")
    torch.cuda.synchronize()
    for (t0, e0), (t1, e1) in zip(marks, marks[1:]):
        acc[t1] = acc.get(t1, 0.0) + e0.elapsed_time(e1)
print({k: round(v / N, 3) for k, v in acc.items()}, "sum", round(sum(acc.values()) / N, 3))
