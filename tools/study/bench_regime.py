"""What state should bench.py time the step at?  Seeded init (rec at O(1e2): every EMD exponential underflows) against the
tests' operating-point recipe (heads x 2^-6, then K untimed training steps of the engine itself on the bench batch)."""
import copy, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-point-clouds-autocomplete_amd")); sys.path.insert(0, os.path.join(ROOT, "tools", "study"))
import bench
from hyperpocket_amd import ops
from hyperpocket_amd.core.engine import TrainEngine
from hyperpocket_amd.core.setup import weights_init
from hyperpocket_amd.model.full_model import FullModel
from emd_cull_check import timeit, lib

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for scale_log2, K in ((None, 0), (-6, 100), (-6, 400), (-6, 800)):
    torch.manual_seed(2020)
    model = FullModel(copy.deepcopy(bench.MODEL_CFG)); model.apply(weights_init); model = model.to(dev)
    if scale_log2 is not None:
        with torch.no_grad():
            for head in model.hyper_network.output:
                head.weight.mul_(2.0 ** scale_log2)
    torch.manual_seed(2020)
    eng = TrainEngine(model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, loss_coef=0.05, emd_coef=0.05)
    ex, mi, gt = bench.synth_batch(64, 1024, dev, 2020)
    for _ in range(K):
        out = eng.step(ex, mi, gt, epoch=1)
    eng.finish_pending(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        out = eng.step(ex, mi, gt, epoch=1)
    eng.finish_pending(); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    model.train()
    with torch.no_grad():
        rec, _, _ = model(ex.clone(), mi.clone(), [64, 2048, 3], 1, dev)
    ops.clear_grad_views()
    rec = rec.permute(0, 2, 1).contiguous()
    print(f"heads x 2^{scale_log2}, {K} steps: step {ms:.3f} ms  loss {float(out['loss_all']):.4g}  rec std {rec.std().item():.3f} min {rec.min().item():.2f} max {rec.max().item():.2f}"
          f" per-cloud std {rec.std(dim=1).mean().item():.3f}  hp_emd_forward(gt, rec): cull0 {[lib.hp_emd_set_cull(0), timeit(gt, rec)][1]:.3f} ms, cull3 {[lib.hp_emd_set_cull(3), timeit(gt, rec)][1]:.3f} ms", flush=True)
