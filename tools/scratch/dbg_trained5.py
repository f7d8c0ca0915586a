import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "."); sys.path.insert(0, "3d-point-clouds-autocomplete_amd")
import numpy as np, torch
from conftest import golden, fixture_state_, OracleLib
import test_model_gpu as T
from hyperpocket_amd._lib import load_library
from hyperpocket_amd.losses.champfer_loss import ChamferLoss
from oracle import hyperpocket_ref as ref
g = golden("model_trained")
lib = load_library()
P = fixture_state_(ref.init_params(int(g["seed"])), g)
t = lambda n: torch.from_numpy(g[n]).double()
def oracle(dtype):
    Pd = {k: v.to(dtype) for k, v in P.items()}
    c = lambda n: t(n).to(dtype)
    z, mu, ex = ref.encoder_forward(Pd, "random_encoder", c("missing"), True, c("eps"))
    rm = ref.encoder_forward(Pd, "real_encoder", c("existing"), False)
    latent = torch.cat([z, rm], 1)
    theta = ref.hypernet_forward(Pd, latent).detach().requires_grad_(True)
    rec = torch.stack([ref.target_forward(theta[b], c("points")[b]) for b in range(4)])    # (B,N,3)
    rec.retain_grad()
    loss = 0.05 * ref.chamfer_loss(c("gt"), rec)
    loss.backward()
    return theta.grad, rec.grad, theta.detach(), rec.detach()
dth64, grec64, th64, rec64 = oracle(torch.float64)
dth32, grec32, th32, rec32 = oracle(torch.float32)
def rel(a, b): return ((a.double().cpu() - b.double()).abs().max() / b.double().abs().max()).item()
print("fp32 torch oracle vs fp64: dtheta", rel(dth32, dth64), "grec", rel(grec32, grec64), "theta", rel(th32, th64))
sl = {"h0": (0, 128), "h1": (128, 2240), "h2": (2240, 10560), "h3": (10560, 18816), "h4": (18816, 19011)}
for tag, off in (("default", None), ("conv_split off", "hp_conv_split_set")):
    was = getattr(lib, off)(0) if off else None
    model = T.build_model(int(g["seed"])); fixture_state_(model.state_dict(), g); model.train()
    keep = {}
    hn_fwd = model.hyper_network.forward
    def fwd(x, _f=hn_fwd):
        th = _f(x); th.retain_grad(); keep["theta"] = th; return th
    model.hyper_network.forward = fwd
    ex, mi, gt = (torch.from_numpy(g[k]).cuda() for k in ("existing", "missing", "gt"))
    rec, lv, mu = model(ex, mi, list(gt.shape), int(g["epoch"]), torch.device("cuda"), points=torch.from_numpy(g["points"]).cuda(), eps=torch.from_numpy(g["eps"]).cuda())
    rec.retain_grad()
    loss_r = torch.mean(0.05 * ChamferLoss().cuda()(gt, rec.permute(0, 2, 1)))
    loss_r.backward()
    th = keep["theta"]
    print(tag, ": theta", rel(th.detach(), th64), "grec", rel(rec.grad.permute(0, 2, 1), grec64), "dtheta", rel(th.grad, dth64),
          {k: f"{rel(th.grad[:, a:b], dth64[:, a:b]):.1e}" for k, (a, b) in sl.items()})
    e = (th.grad.cpu().double() - dth64).abs()
    bi = np.unravel_index(e.argmax().item(), e.shape)
    print("   worst at", bi, "got", th.grad[bi].item(), "want", dth64[bi].item(), "count > 1e-4*max:", (e > 1e-4 * dth64.abs().max()).sum().item())
    if off: getattr(lib, off)(was)
