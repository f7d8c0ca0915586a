#!/usr/bin/env python3
"""Runs the engine at a few other shapes (BASELINE config 5 per-GPU shape, ragged N, B=1, B not a power of two): a crash check, not a parity test."""
import copy, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-point-clouds-autocomplete_amd"))
import bench
from hyperpocket_amd.core.setup import weights_init
from hyperpocket_amd.model.full_model import FullModel
from hyperpocket_amd.core.engine import TrainEngine
torch.manual_seed(0)
m = FullModel(copy.deepcopy(bench.MODEL_CFG)); m.apply(weights_init); m = m.cuda()
for B, N, emd in [(8, 8192, 0.0), (8, 8192, 0.05), (64, 8192, 0.0), (3, 1000, 0.05), (1, 2048, 0.05), (130, 512, 0.05)]:
    eng = TrainEngine(m, emd_coef=emd)
    ex, mi, gt = bench.synth_batch(B, N // 2, torch.device("cuda"), 1)
    for _ in range(2): out = eng.step(ex, mi, gt, 1)
    eng.finish_pending(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): out = eng.step(ex, mi, gt, 1)
    eng.finish_pending(); torch.cuda.synchronize()
    print(f"B={B} N={N} emd={emd}: {(time.perf_counter()-t0)/3*1e3:.2f} ms/step loss {out['loss_all'].item():.4g}", flush=True)
    from hyperpocket_amd import ops; ops.clear_grad_views()
