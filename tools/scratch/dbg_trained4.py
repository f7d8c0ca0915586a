import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "."); sys.path.insert(0, "3d-point-clouds-autocomplete_amd")
import numpy as np, torch
from conftest import golden, fixture_state_, OracleLib
import test_model_gpu as T
from hyperpocket_amd._lib import load_library
g = golden("model_trained")
lib = load_library(); ol = OracleLib()
rec_ref = np.ascontiguousarray(g["rec"].transpose(0, 2, 1))
d1r, i1r, d2r, i2r = ol.nndistance(g["gt"], rec_ref)
for tag, off in (("default", None), ("conv_split off", "hp_conv_split_set")):
    was = getattr(lib, off)(0) if off else None
    model = T.build_model(int(g["seed"])); fixture_state_(model.state_dict(), g); model.train()
    ex, mi, gt = (torch.from_numpy(g[k]).cuda() for k in ("existing", "missing", "gt"))
    with torch.no_grad():
        rec, lv, mu = model(ex, mi, list(gt.shape), int(g["epoch"]), torch.device("cuda"), points=torch.from_numpy(g["points"]).cuda(), eps=torch.from_numpy(g["eps"]).cuda())
    r = rec.permute(0, 2, 1).contiguous().cpu().numpy()
    d1, i1, d2, i2 = ol.nndistance(g["gt"], r)
    print(tag, "idx1 flips", np.argwhere(i1 != i1r).tolist(), "idx2 flips", np.argwhere(i2 != i2r).tolist())
    for (b, j) in np.argwhere(i2 != i2r):
        # margins at the flipped point
        dd = ((g["gt"][b] - rec_ref[b, j]) ** 2).sum(1); s = np.sort(dd)
        print("   rec point", b, j, "best two d:", s[0], s[1], "gap", s[1] - s[0])
    for (b, j) in np.argwhere(i1 != i1r):
        dd = ((rec_ref[b] - g["gt"][b, j]) ** 2).sum(1); s = np.sort(dd)
        print("   gt point", b, j, "best two d:", s[0], s[1], "gap", s[1] - s[0])
    if off: getattr(lib, off)(was)
