import sys, os, contextlib
sys.path.insert(0, "tests"); sys.path.insert(0, "."); sys.path.insert(0, "3d-point-clouds-autocomplete_amd")
import numpy as np, torch
from conftest import golden, fixture_state_
import test_model_gpu as T
from hyperpocket_amd import ops
from hyperpocket_amd.losses.champfer_loss import ChamferLoss
g = golden("model_trained")
for arith in ("default", "strict"):
    with (ops.strict_fp32() if arith == "strict" else contextlib.nullcontext()):
        model = T.build_model(int(g["seed"]))
        fixture_state_(model.state_dict(), g)
        model.train()
        ex, mi, gt = (torch.from_numpy(g[k]).cuda() for k in ("existing", "missing", "gt"))
        rec, lv, mu = model(ex, mi, list(gt.shape), int(g["epoch"]), torch.device("cuda"), points=torch.from_numpy(g["points"]).cuda(), eps=torch.from_numpy(g["eps"]).cuda())
        print(arith, "rec err", (rec.cpu() - torch.from_numpy(g["rec"])).abs().max().item())
        loss_r = torch.mean(0.05 * ChamferLoss().cuda()(gt, rec.permute(0, 2, 1)))
        kld = 0.5 * (torch.exp(lv) + mu * mu - 1 - lv).sum() / 4
        (loss_r + kld).backward()
        for k, p in model.named_parameters():
            key = k.replace(".", "__")
            if "gnone__" + key in g: continue
            if "gfull__" + key in g:
                a, b = p.grad.flatten().cpu().double(), torch.from_numpy(g["gfull__" + key]).double()
            else:
                a, b = p.grad.flatten()[torch.from_numpy(g["gidx__" + key]).cuda()].cpu().double(), torch.from_numpy(g["gsamp__" + key]).double()
            e = (a - b).abs().max().item(); sc = b.abs().max().item()
            if e > 1e-4 * sc:
                print(f"  {k}: err {e:.3e} scale {sc:.3e} rel {e/sc:.2e}")
