"""A/B of the fused encoder backward's delta chain: f16 pipe (enc_bwd_f16.hip) vs fp32 MFMA chain, HyperPocket shape
(two encoders, B=64, Np=1024).  Same process, interleaved, HIP events around the pair backward."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "3d-point-clouds-autocomplete_amd"))
from hyperpocket_amd import _lib
from hyperpocket_amd.model.encoder import Encoder
from hyperpocket_amd.core.setup import weights_init

lib = _lib.load_library()
torch.manual_seed(0)
B, Np = 64, 1024
encs = [Encoder({"output_size": 128, "use_bias": True, "relu_slope": 0.2}, is_vae=v).apply(weights_init).cuda() for v in (True, False)]
xs = [(torch.rand(B, Np, 3, device="cuda") - 0.5).transpose(1, 2) for _ in range(2)]
eps = torch.randn(B, 128, device="cuda")

def step():
    for e in encs:
        for p in e.parameters():
            p.grad = None
    o0 = encs[0](xs[0], eps)
    o1 = encs[1](xs[1])
    (sum(o.sum() for o in o0) + o1.sum()).backward()

def timed(on, n=30):
    prev = lib.hp_encoder_backward_set_chain_f16(on)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    lib.hp_encoder_backward_set_chain_f16(prev)
    return dt

for rep in range(3):
    print("chain f16 %.4f ms   fp32 chain %.4f ms (forward + backward of both encoders, separate launches)" % (timed(1), timed(0)))
