"""Loss trajectories of the engine from the bench's pre-conditioning start with hp_emd_set_cull(0) and (3): must agree to rounding
at first and drift apart only slowly (chaotic amplification), not systematically."""
import copy, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-point-clouds-autocomplete_amd"))
import bench
from hyperpocket_amd._lib import load_library
from hyperpocket_amd.core.engine import TrainEngine
from hyperpocket_amd.core.setup import weights_init
from hyperpocket_amd.model.full_model import FullModel
lib = load_library()
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
traj = {}
for K in (0, 3, 0):
    lib.hp_emd_set_cull(K)
    torch.manual_seed(2020)
    model = FullModel(copy.deepcopy(bench.MODEL_CFG)); model.apply(weights_init); model = model.to(dev)
    with torch.no_grad():
        for head in model.hyper_network.output:
            head.weight.mul_(2.0 ** -6)
    torch.manual_seed(2020)
    eng = TrainEngine(model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, loss_coef=0.05, emd_coef=0.05)
    ex, mi, gt = bench.synth_batch(64, 1024, dev, 2020)
    ls = []
    for s in range(460):
        out = eng.step(ex, mi, gt, epoch=1)
        if s < 12 or s % 40 == 0 or s > 445:
            ls.append((s, float(out["loss_all"]), float(out.get("loss_emd", float("nan"))) if isinstance(out, dict) else 0))
    eng.finish_pending()
    traj.setdefault(K, []).append(ls)
print("keys of step output:", list(out.keys()))
a, b, a2 = traj[0][0], traj[3][0], traj[0][1]
for (s, la, ea), (_, lb, eb), (_, la2, _) in zip(a, b, a2):
    print(f"step {s:4d}: cull0 {la:.6f}  cull3 {lb:.6f}  rel diff {abs(la-lb)/abs(la):.2e}   (cull0 again: {la2:.6f})")
