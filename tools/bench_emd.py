import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-point-clouds-autocomplete_amd"))
from hyperpocket_amd.utils.pytorch_structural_losses.match_cost import match_cost
b, n = 64, 2048
x = torch.rand(b, n, 3, device="cuda") - 0.5
y = (torch.rand(b, n, 3, device="cuda") - 0.5).requires_grad_(True)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    y.grad = None
    match_cost(x, y).sum().backward()
torch.cuda.synchronize()
print("done")
