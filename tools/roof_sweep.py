import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "3d-point-clouds-autocomplete_amd"))
import torch, bench
from hyperpocket_amd.ops import gemm
torch.cuda.set_device(0)
m = 64 * 1024
a = torch.randn(m, 512, device="cuda"); w = torch.randn(512, 512, device="cuda") * 0.05; b = torch.zeros(512, device="cuda"); c = torch.empty(m, 512, device="cuda")
fl = 2.0 * m * 512 * 512
for warm, iters in ((20, 40), (100, 40), (400, 40), (400, 400), (2000, 400), (20, 40)):
    ms = bench.event_time_ms(lambda: gemm(a, w, bias=b, out=c), iters=iters, warm=warm)
    print(warm, iters, round(ms * 1e3, 1), "us", round(fl / ms / 1e9, 1), "TF", flush=True)
