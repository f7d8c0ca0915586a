import sys, os
sys.path.insert(0, "tools/study"); sys.path.insert(0, "tools")
from emd_cull_check import lib, emd
from emd_cull_share import regimes
import torch
name, (gt, rec) = list(regimes(per=64).items())[1]
lib.hp_emd_set_cull(3)
for it in range(3):
    keep = {}
    emd(gt, rec, False, True, keep)
    per = lib.hp_approxmatch_workspace_floats(1, 2048, 2048)
    w = keep["ws"].view(64, per)
    print(w[:, per - 16:per - 10].cpu().numpy()[[0, 1, 31, 32, 63]] * 0.01, "us: flag, load, level0, all levels, records, boxes")
