import sys, os, contextlib
sys.path.insert(0, "tests"); sys.path.insert(0, "."); sys.path.insert(0, "3d-point-clouds-autocomplete_amd")
import numpy as np, torch
from conftest import golden, fixture_state_
import test_model_gpu as T
from hyperpocket_amd import ops
from hyperpocket_amd._lib import load_library
from oracle import hyperpocket_ref as ref
g = golden("model_trained")
lib = load_library()
P = fixture_state_(ref.init_params(int(g["seed"])), g)
Pd = {k: v.double() for k, v in P.items()}
t = lambda n: torch.from_numpy(g[n]).double()
z64, mu64, ex64 = ref.encoder_forward(Pd, "random_encoder", t("missing"), True, t("eps"))
rm64 = ref.encoder_forward(Pd, "real_encoder", t("existing"), False)
z32, mu32, ex32 = ref.encoder_forward(P, "random_encoder", t("missing").float(), True, t("eps").float())
rm32 = ref.encoder_forward(P, "real_encoder", t("existing").float(), False)
print("fp64: |z| max", z64.abs().max().item(), "|mu|", mu64.abs().max().item(), "explv max", ex64.max().item(), "real_mu", rm64.abs().max().item())
def rel(a, b): return ((a.double().cpu() - b).abs().max() / b.abs().max()).item()
print("torch fp32 oracle: z", rel(z32, z64), "mu", rel(mu32, mu64), "explv", rel(ex32, ex64), "real_mu", rel(rm32, rm64))
for tag, off in (("default", None), ("conv_split off", "hp_conv_split_set"), ("presplit off", "hp_conv_presplit_set")):
    was = getattr(lib, off)(0) if off else None
    model = T.build_model(int(g["seed"])); fixture_state_(model.state_dict(), g); model.train()
    with torch.no_grad():
        z, mu, ex = model.random_encoder(torch.from_numpy(g["missing"]).cuda().transpose(1, 2), torch.from_numpy(g["eps"]).cuda())
        rm = model.real_encoder(torch.from_numpy(g["existing"]).cuda().transpose(1, 2))
    print(tag, ": z", rel(z, z64), "mu", rel(mu, mu64), "explv", rel(ex, ex64), "real_mu", rel(rm, rm64))
    if off: getattr(lib, off)(was)
# layer-wise magnitude of the fp64 activations (outliers?)
h = t("missing")
for i, li in enumerate((0, 2, 4, 6, 8)):
    h = h @ Pd[f"random_encoder.conv.{li}.weight"][:, :, 0].t() + Pd[f"random_encoder.conv.{li}.bias"]
    if i < 4: h = torch.relu(h)
    print("layer", i + 1, "absmax", h.abs().max().item(), "rms", h.pow(2).mean().sqrt().item(), "bias absmax", Pd[f"random_encoder.conv.{li}.bias"].abs().max().item())
gp = h.max(dim=1)[0]
print("pooled absmax", gp.abs().max().item(), "min", gp.min().item())
