#!/usr/bin/env python3
"""GPU box: randomised shapes / scales through hp_encoder_forward with the split-f16 conv stack and with the fp32 MFMA GEMMs —
the pooled features must agree to fp32 rounding and the arg-max rows must attain the fp64 maximum within the same bound."""
import os, sys, ctypes, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-point-clouds-autocomplete_amd"))
import torch
from hyperpocket_amd import _lib, ops
from hyperpocket_amd.model.encoder import Encoder
from hyperpocket_amd.core.setup import weights_init

lib = _lib.load_library()
random.seed(7)
torch.manual_seed(7)
worst = 0.0
for trial in range(60):
    B = random.choice([1, 2, 3, 7, 16, 33, 64])
    Np = random.choice([1, 2, 5, 31, 127, 128, 129, 255, 256, 300, 1000, 1024, 2048])
    if B * Np > 64 * 1024:
        continue
    xs, ws = 10 ** random.uniform(-4, 3), 10 ** random.uniform(-1, 0.7)
    enc = Encoder({"output_size": 128, "use_bias": True, "relu_slope": 0.2}, is_vae=False).apply(weights_init).cuda()
    params = [p.detach().reshape(p.shape[0], -1).contiguous() if p.dim() == 3 else p.detach().contiguous() for p in enc._params()]
    with torch.no_grad():
        for i in range(5):
            params[i].mul_(ws)
            torch.nn.init.uniform_(params[5 + i], -0.05 * ws, 0.05 * ws)
    x = ((torch.rand(B, Np, 3, device="cuda") - 0.5) * xs).contiguous()
    out = {}
    for split in (1, 0):
        f32 = dict(dtype=torch.float32, device="cuda")
        argidx = torch.empty((B, 512), dtype=torch.int32, device="cuda")
        g, f, mu = torch.empty((B, 512), **f32), torch.empty((B, 512), **f32), torch.empty((B, 128), **f32)
        wsb = torch.empty((ops._long_fn("hp_encoder_forward_workspace_floats", B, Np),), **f32)
        w = ops._encoder_struct(params)
        prev = lib.hp_conv_split_set(split)
        _lib.call("hp_encoder_forward", B, Np, x, ctypes.byref(w), 128, 0, None, argidx, g, f, mu, None, None, None, wsb,
                  _lib.current_stream(x.device))
        torch.cuda.synchronize()
        lib.hp_conv_split_set(prev)
        out[split] = (g.clone(), argidx.clone(), mu.clone())
    h = x.view(B * Np, 3).double()
    for l in range(4):
        h = torch.relu(h @ params[l].double().t() + params[5 + l].double())
    h5 = (h @ params[4].double().t() + params[9].double()).view(B, Np, 512)
    want_g = h5.max(dim=1).values
    scale = max(want_g.abs().max().item(), 1e-30)
    for split in (1, 0):
        g, arg, mu = out[split]
        assert torch.isfinite(g).all() and torch.isfinite(mu).all(), (trial, B, Np, split)
        assert (arg >= 0).all() and (arg < Np).all(), (trial, B, Np, split)
        eg = (g.double() - want_g).abs().max().item() / scale
        at = torch.gather(h5, 1, arg.long().unsqueeze(1)).squeeze(1)
        ea = (at - want_g).abs().max().item() / scale
        worst = max(worst, eg, ea) if split else worst
        assert eg <= 5e-6 and ea <= 1e-5, (trial, B, Np, xs, ws, split, eg, ea)
    print(f"trial {trial:2d}  B={B:2d} Np={Np:4d} xscale={xs:8.2e} wscale={ws:5.2f}  ok", flush=True)
print("worst relative error of the split path vs the fp64 chain:", worst)
