import sys, os, copy, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-point-clouds-autocomplete_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_dp_gpu as T
from hyperpocket_amd import ops, _lib
from hyperpocket_amd.core.engine import TrainEngine
ex, mi, gt, pts, eps = (t.cuda() for t in T._data())
for fused in (1, 0):
    _lib.load_library().hp_encoder_backward_set_fused(fused)
    res = []
    for rep in range(3):
        model = T._build()
        eng = TrainEngine(model, emd_coef=0.05, fuse_heads_adam=(rep != 2))
        for _ in range(3):
            eng.step(ex, mi, gt, 7, points=pts, eps_noise=eps)
        eng.synchronize()
        res.append({k: p.detach().clone() for k, p in model.named_parameters()})
        eng.close(); ops.clear_grad_views()
    for j in (1, 2):
        bad = [k for k in res[0] if not torch.equal(res[0][k], res[j][k])]
        print("fused", fused, "run0 vs run", j, "(fuse_heads_adam", j != 2, "): differing params:", len(bad), bad[:6])
