import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/3d-point-clouds-autocomplete_amd")
import bench
from hyperpocket_amd.ops import gemm
m=65536
a = torch.randn(m, 512, device="cuda"); w = torch.randn(512, 512, device="cuda") * 0.05; b = torch.zeros(512, device="cuda")
c = torch.empty(m, 512, device="cuda")
print("alloc", bench.event_time_ms(lambda: gemm(a, w, bias=b), iters=20, warm=3))
print("prealloc", bench.event_time_ms(lambda: gemm(a, w, bias=b, out=c), iters=20, warm=3))
