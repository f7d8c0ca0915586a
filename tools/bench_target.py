import copy, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-point-clouds-autocomplete_amd"))
from hyperpocket_amd.model.target_network import target_network_batched
from hyperpocket_amd import ops
cfg = {"use_bias": True, "layer_out_channels": [32, 64, 128, 64]}
B = 64
theta = (torch.randn(B, 19011, device="cuda") * 0.1).requires_grad_(True)
pts = ops.sample_points(B, 2048, 0.0, 1, 1, "cuda")
gy = torch.randn(B, 2048, 3, device="cuda")
for _ in range(5):
    theta.grad = None
    target_network_batched(cfg, theta, pts).backward(gy)
torch.cuda.synchronize()
