"""GPU: the data-parallel TrainEngine end to end with 2 ranks (both on cuda:0, gloo as the transport so that one GPU
suffices; the production transport is RCCL, same torch.distributed calls).  Two ranks, 2 clouds each, must end a
step with the same parameters as one process stepping on all 4 clouds (SUM of Chamfer/EMD gradients, KLD over the
GLOBAL batch — SURVEY §8e)."""
import copy
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

CFG = {
    "random_encoder": {"output_size": 128, "use_bias": True, "relu_slope": 0.2},
    "real_encoder": {"output_size": 128, "use_bias": True, "relu_slope": 0.2},
    "hyper_network": {"use_bias": True, "relu_slope": 0.2},
    "target_network": {"use_bias": True, "relu_slope": 0.2, "freeze_layers_learning": False, "layer_out_channels": [32, 64, 128, 64]},
    "target_network_input": {"constant": False, "normalization": {"enable": True, "type": "progressive", "epoch": 100}},
}


def _data(n=4):
    g = torch.Generator().manual_seed(3)
    ex, mi = torch.rand(n, 128, 3, generator=g) - 0.5, torch.rand(n, 128, 3, generator=g) - 0.5
    pts, eps = torch.rand(n, 256, 3, generator=g) * 2 - 1, torch.randn(n, 128, generator=g)
    return ex, mi, torch.cat([ex, mi], 1), pts, eps


def _build():
    from hyperpocket_amd.core.setup import weights_init
    from hyperpocket_amd.model.full_model import FullModel
    torch.manual_seed(99)
    m = FullModel(copy.deepcopy(CFG))
    m.apply(weights_init)
    return m.cuda()


def _worker(rank, world, port, out, shard, nclouds=4):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "3d-point-clouds-autocomplete_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from hyperpocket_amd.core.engine import TrainEngine
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    model = _build()
    if rank != 0:                      # replicas must start from rank 0's weights: perturb, the engine's broadcast repairs it
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.01)
    eng = TrainEngine(model, emd_coef=0.05, shard_heads=shard)
    assert (eng.shard is not None) == shard
    ex, mi, gt, pts, eps = (t.cuda() for t in _data(nclouds))
    per = nclouds // world
    sl = slice(rank * per, rank * per + per)
    for _ in range(2):
        res = eng.step(ex[sl].contiguous(), mi[sl].contiguous(), gt[sl].contiguous(), 7, points=pts[sl].contiguous(),
                       eps_noise=eps[sl].contiguous())
    eng.finish_pending()       # the heads' exchange/update is deferred into the next step; flush it before reading
    torch.cuda.synchronize()
    osd = eng.optimizer_state_dict()     # collective: gathers the row-sharded moments of the heads
    if rank == 0:   # by file: a 173 MB dict does not travel well through an mp.Queue once the sender exits
        torch.save({"params": {k: p.detach().cpu() for k, p in model.named_parameters()},
                    "opt": {i: {k: v.cpu() for k, v in st.items()} for i, st in osd["state"].items()}}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("shard,world,nclouds", [(True, 2, 4), (False, 2, 4), (True, 8, 8)],
                         ids=["sharded-heads", "all-reduce", "eight-ranks-sharded-heads"])
def test_two_rank_engine_matches_single_process_global_batch(shard, world, nclouds):
    """Both exchanges of the heads' gradient: `shard` = ranks all-gather d theta / t5, each updates its row slice of the
    heads and the updated rows are all-gathered (HeadsShard); otherwise the flat gradient is all-reduced.
    `eight-ranks`: the node's world size — 8 ranks (all on cuda:0, gloo), one cloud each: the W = 8 row sharding of the
    19011 head rows (2377-row slices, the last one ragged: the 19016-row padding), the deferred gathers and the small-bucket
    all-reduce run end to end with the real kernels, against one process stepping on the 8 clouds."""
    from hyperpocket_amd import ops
    from hyperpocket_amd.core.engine import TrainEngine
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    import tempfile
    out = os.path.join(tempfile.mkdtemp(), "rank0_params.pt")
    procs = [ctx.Process(target=_worker, args=(r, world, port, out, shard, nclouds)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=900)
        assert p.exitcode == 0
    saved = torch.load(out)
    got, got_opt = saved["params"], saved["opt"]
    os.remove(out)
    model = _build()
    eng = TrainEngine(model, emd_coef=0.05)
    try:
        ex, mi, gt, pts, eps = (t.cuda() for t in _data(nclouds))
        for _ in range(2):
            eng.step(ex, mi, gt, 7, points=pts, eps_noise=eps)
        eng.finish_pending()
        for k, p in model.named_parameters():
            a, b = got[k].double(), p.detach().cpu().double()
            # Adam's first updates are ~lr*sign(g) (lr = 1e-4): an element whose gradient is rounding noise may take the
            # other sign under a different summation order (two ranks of 4 clouds vs one process of 8) — at most
            # 2*lr per step, 4e-4 after the two steps — and only a handful of elements may do so
            d = (a - b).abs()
            assert d.max().item() <= 4.1e-4, k
            assert (d > 1e-5).double().mean().item() <= 2e-3, k
            assert d.mean().item() <= 2e-6, k
        # the optimiser checkpoint written under DP (heads' Adam moments row-sharded over the ranks, gathered by
        # optimizer_state_dict) equals the single-process optimiser state on the global batch
        want_opt = eng.optimizer_state_dict()["state"]
        assert sorted(got_opt) == sorted(want_opt)
        for i, st in want_opt.items():
            assert float(got_opt[i]["step"]) == float(st["step"]) == 2.0
            for key in ("exp_avg", "exp_avg_sq"):
                a, b = got_opt[i][key].double(), st[key].cpu().double()
                assert (a - b).abs().max().item() <= 2e-4 * b.abs().max().item() + 1e-12, (i, key)
    finally:
        ops.clear_grad_views()


def test_heads_dw_rows_equals_slice_of_full_product():
    """hp_hypernet_heads_dw_rows: rows [r0, r0+rows) of dtheta_all^T . t5_all, at the 8-rank shape (Kc = 8*64) and at a
    ragged last slice."""
    import ctypes
    from hyperpocket_amd._lib import call, current_stream, load_library
    lib = load_library()
    lib.hp_hypernet_heads_dw_workspace_floats.restype = ctypes.c_long
    g = torch.Generator().manual_seed(1)
    for kc, r0, rows in ((512, 2377 * 3, 2377), (128, 19016 // 2, 19011 - 19016 // 2), (4, 0, 130)):
        dth = torch.randn(kc, 19011, generator=g).cuda()
        t5 = torch.randn(kc, 2048, generator=g).cuda()
        out = torch.full((rows, 2048), float("nan"), device="cuda")
        ws = torch.empty(lib.hp_hypernet_heads_dw_workspace_floats(), device="cuda")
        call("hp_hypernet_heads_dw_rows", kc, rows, r0, dth, 19011, t5, out, ws, current_stream(out.device))
        want = dth[:, r0:r0 + rows].double().t() @ t5.double()
        err = (out.double() - want).abs().max().item()
        assert err <= 2e-5 * want.abs().max().item(), (kc, r0, rows, err)


def test_bench_runs_every_collective_in_a_one_rank_rccl_group():
    """bench.py with HP_BENCH_FORCE_EXCHANGE=1: a one-rank RCCL (backend "nccl") group in which the multi-rank step
    really issues its collectives — broadcast, the d theta / t5 gathers, the in-place gather of the updated heads rows,
    both all-reduces, the deferred waits, the barrier — on the production transport."""
    import json
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HP_BENCH_FORCE_EXCHANGE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "2", "--batch", "8",
                        "--no-extras", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["final_loss"] == line["final_loss"]


def _bench(args, extra_env, timeout=900):
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_gpus_2_starts_two_ranks_itself():
    """`python bench.py --gpus 2` with no torchrun around it: bench.py spawns the two ranks (before touching the GPU) and
    rank 0 prints ONE line with n_gpus = 2.  One GPU here, so both ranks sit on cuda:0 and talk over gloo (the test hooks
    HP_BENCH_BACKEND / HP_BENCH_ONE_DEVICE); on a multi-GPU node the same command line runs one rank per GPU over RCCL."""
    line = _bench(["--gpus", "2", "--steps", "3", "--warmup", "2", "--batch", "8", "--no-extras", "--no-cpu-baseline"],
                  {"HP_BENCH_BACKEND": "gloo", "HP_BENCH_ONE_DEVICE": "1"})
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2
    assert line["config"]["global_batch"] == 16 and line["value"] > 0 and line["final_loss"] == line["final_loss"]


def test_bench_chamfer_stress_workload_line():
    """BASELINE.json configs[4] at a reduced batch (the bench default is B=64/GPU, N=8192): one line with pairs/s and the
    VALU roofline object."""
    line = _bench(["--workload", "chamfer-stress", "--batch", "4", "--steps", "3", "--warmup", "1"], {})
    assert line["unit"] == "pairs/s" and line["config"]["points"] == 8192 and line["n_gpus"] == 1
    assert abs(line["value"] - 2 * 4 * 8192 * 8192 / (line["ms_per_step"] * 1e-3)) <= 1e-3 * line["value"]
    rf = line["roofline"]
    assert rf["bound"] == "valu" and 0 < rf["frac"] < 1 and rf["flops_per_launch"] == 8.0 * 2 * 4 * 8192 * 8192


def test_fused_heads_dw_adam_equals_dw_then_adam():
    """hp_hypernet_heads_dw_adam (gradient tile in registers -> Adam in place) against the two-pass form it replaces:
    hp_hypernet_heads_dw_rows into a gradient buffer, then hp_adam_step over it.  Row slices as the ranks own them."""
    import ctypes
    from hyperpocket_amd import ops
    from hyperpocket_amd._lib import call, current_stream, load_library
    lib = load_library()
    lib.hp_hypernet_heads_dw_workspace_floats.restype = ctypes.c_long
    g = torch.Generator().manual_seed(5)
    for kc, r0, rows, step in ((64, 0, 19011, 1), (128, 2377 * 5, 2377, 7), (6, 19000, 11, 3)):
        dth = (torch.randn(kc, 19011, generator=g) * 0.1).cuda()
        t5 = torch.randn(kc, 2048, generator=g).cuda()
        w = torch.randn(rows, 2048, generator=g).cuda()
        m = (torch.randn(rows, 2048, generator=g) * 0.01).cuda()
        v = (torch.rand(rows, 2048, generator=g) * 1e-3).cuda()
        w2, m2, v2 = w.clone(), m.clone(), v.clone()
        grad = torch.empty(rows, 2048, device="cuda")
        ws = torch.empty(lib.hp_hypernet_heads_dw_workspace_floats(), device="cuda")
        call("hp_hypernet_heads_dw_rows", kc, rows, r0, dth, 19011, t5, grad, ws, current_stream(w.device))
        ops.adam_step(w2.view(-1), grad.view(-1), m2.view(-1), v2.view(-1), 1e-4, 0.9, 0.999, 1e-8, step)
        call("hp_hypernet_heads_dw_adam", kc, rows, r0, dth, 19011, t5, w, m, v, 1e-4, 0.9, 0.999, 1e-8, step,
             current_stream(w.device))
        assert torch.equal(w, w2) and torch.equal(m, m2) and torch.equal(v, v2), (kc, r0, rows)


def _one_rank_rccl_worker(port, out):
    """A one-rank RCCL group in which every collective of the multi-rank step really runs: Chamfer+EMD steps with the
    row-sharded heads update (factor all-gathers, hp_hypernet_heads_dw_adam on the rank's rows, IN-PLACE all-gather of the
    updated rows: input aliases output) and with the flat all-reduce."""
    import torch.distributed as dist
    from hyperpocket_amd import ops
    from hyperpocket_amd.core.engine import TrainEngine
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    ex, mi, gt, pts, eps = (t.cuda() for t in _data())
    res = {}
    for shard in (True, False):
        model = _build()
        eng = TrainEngine(model, emd_coef=0.05, force_exchange=True, shard_heads=shard)
        assert eng.exchange and (eng.shard is not None) == shard
        snaps = []
        for _ in range(3):
            eng.step(ex, mi, gt, 7, points=pts, eps_noise=eps)
            eng.synchronize()
            snaps.append({k: p.detach().cpu().clone() for k, p in model.named_parameters()})
        res[shard] = snaps
        eng.close()
        ops.clear_grad_views()
    torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


def test_one_rank_rccl_sharded_heads_equal_all_reduce():
    """After the FIRST step the heads' weights of the two exchange modes are bit-identical (same d theta, same t5; the rows'
    fused dW + Adam kernel equals dW-then-Adam bit for bit, and the in-place all-gather of the updated rows must not
    disturb them); everything else — reached through d t5, whose GEMM is split differently when the heads' dW is left to
    the exchange — and the later steps agree within Adam's sign-like sensitivity."""
    import tempfile
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = os.path.join(tempfile.mkdtemp(), "res.pt")
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_one_rank_rccl_worker, args=(port, out))
    p.start()
    p.join(timeout=600)
    assert p.exitcode == 0
    res = torch.load(out)
    os.remove(out)
    a, b = res[True], res[False]
    for k in a[0]:
        if k.startswith("hyper_network.output") and k.endswith("weight"):
            assert torch.equal(a[0][k], b[0][k]), k
    for step in range(3):
        for k in a[step]:
            d = (a[step][k].double() - b[step][k].double()).abs()
            assert d.max().item() <= 2.05e-4 * (step + 1), (step, k, d.max().item())
            assert (d > 1e-5).double().mean().item() <= 2e-3, (step, k)
            assert d.mean().item() <= 2e-6, (step, k)
