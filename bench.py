#!/usr/bin/env python3
"""bench.py — HyperPocket training-step throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--no-emd] [--no-cpu-baseline]

A "step" is one full training step of the hot path on one batch of synthetic clouds resident in
HBM: forward (2 encoders -> hypernetwork -> batched target networks on freshly sampled points)
-> 0.05*Chamfer + KLD/B + 0.05*EMD/N -> backward -> [SUM all-reduce of the flat gradient over
RCCL when N>1] -> Adam.  Workload = the configuration the metric is quoted on: HyperPocket 128+128,
B=64 clouds per GPU, existing/missing (B,1024,3), gt (B,2048,3), fp32 ("Chamfer+EMD").
value = clouds/s over all ranks (weak scaling: B per GPU fixed).

One JSON line on rank 0; besides the contract keys it carries
  roofline      the dominant kernel (fp32 MFMA GEMM of the encoder stack), timed live with HIP events
  cpu_baseline  the oracle's torch-CPU restatement of the reference step timed on this box's cores
  breakdown     extra figures (Chamfer-only step, per-op times) — informational
"""
import argparse
import copy
import gc
import json
import os
import sys
import time

# ROCclr multiplexes HIP streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues per priority; streams that share
# one serialise.  The step uses the caller's stream, a side stream and RCCL's stream(s): give each its own queue.
# Must be in the environment before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "3d-point-clouds-autocomplete_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

MODEL_CFG = {
    "random_encoder": {"output_size": 128, "use_bias": True, "relu_slope": 0.2},
    "real_encoder": {"output_size": 128, "use_bias": True, "relu_slope": 0.2},
    "hyper_network": {"use_bias": True, "relu_slope": 0.2},
    "target_network": {"use_bias": True, "relu_slope": 0.2, "freeze_layers_learning": False,
                       "layer_out_channels": [32, 64, 128, 64]},
    "target_network_input": {"constant": False, "normalization": {"enable": True, "type": "progressive", "epoch": 100}},
}
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E spec peak


def synth_batch(b, n_half, device, seed):
    """existing, missing ~ U(-0.5,0.5)^(B,1024,3), gt = cat(existing, missing)."""
    g = torch.Generator(device=device).manual_seed(seed)
    ex = torch.rand(b, n_half, 3, device=device, generator=g) - 0.5
    mi = torch.rand(b, n_half, 3, device=device, generator=g) - 0.5
    return ex, mi, torch.cat([ex, mi], 1)


def event_time_ms(fn, iters, warm=2):
    """Average duration of fn() measured with HIP events on the stream fn launches on (torch's current stream).
    The events bracket back-to-back launches, so a host-side pause longer than the queued work would be counted as
    kernel time: Python's cyclic collector (a 30 ms gen-2 pass was observed here) is held off for the region.  The
    start event is recorded straight behind the warm-up launches, with no synchronisation in between: an idle gap there
    makes the chip re-ramp its clock inside the timed region (measured: 102 vs 113-118 TFLOP/s on the same launches at
    20 warm-up launches)."""
    gc.collect()
    gc.disable()
    try:
        for _ in range(warm):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
    finally:
        gc.enable()
    return s.elapsed_time(e) / iters


def roofline_dominant_kernel(batch, n_half):
    """The kernel with the largest share of the step (profiles/): gemm_kernel<128,128,4,2,16,4>, the fp32-MFMA GEMM
    that runs the encoder's wide layers (M = B*1024 points).  One launch = layer 5 of one encoder:
    C(M x 512) = A(M x 512) W(512 x 512)^T + b.  Algorithmic flops = 2*M*512*512 (868 736 FLOP/point
    of which layer 5 is 2*512*512).  `traffic` (HBM bytes per launch) comes from the PMC passes recorded in
    profiles/r01_pmc_gemm_conv5.json (FETCH_SIZE x2 correction + WRITE_SIZE), measured at B=64."""
    from hyperpocket_amd.ops import gemm
    m = batch * n_half
    a = torch.randn(m, 512, device="cuda")
    w = torch.randn(512, 512, device="cuda") * 0.05
    b = torch.zeros(512, device="cuda")
    c = torch.empty(m, 512, device="cuda")
    # 200 warm-up launches (~55 ms): the chip needs >20 ms of continuous load to reach the clock it then sustains — the
    # state every kernel of a training run executes in (tools/roof_sweep.py: 20 warm-up launches read 112 TFLOP/s, 100,
    # 400 or 2000 read 127 on the same box; the step time itself does not depend on the warm-up length)
    ms = event_time_ms(lambda: gemm(a, w, bias=b, out=c), iters=100, warm=200)
    flops = 2.0 * m * 512 * 512
    achieved = flops / (ms * 1e-3) / 1e12
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r01_pmc_gemm_conv5.json")
    if os.path.exists(pmc) and batch == 64 and n_half == 1024:
        traffic = json.load(open(pmc))["hbm_bytes_per_launch"]
    return {"bound": "mfma", "kernel": "gemm_kernel<128,128,4,2,16,4> (encoder conv5: M=B*1024, N=K=512, fp32 MFMA 32x32x2)",
            "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
            "avg_launch_ms": round(ms, 4), "flops_per_launch": flops,
            "algorithmic_bytes_per_launch": (2 * m * 512 + 512 * 512 + 512) * 4}


def cpu_baseline(n_half, emd_coef, sample_b=4, timed_steps=2):
    """The oracle (oracle/hyperpocket_ref.py: torch-CPU restatement of the reference step + the C restatement of the
    EMD kernels, kind "port") on a bounded sample of the SAME workload: `sample_b` clouds of the same per-cloud shape,
    same loss terms, 1 warm-up + `timed_steps` timed steps."""
    from oracle import hyperpocket_ref as ref
    # torch's CPU kernels stop scaling (and thrash across NUMA domains) far below a 256-core host: 16 threads
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    os.environ["OMP_NUM_THREADS"] = str(threads)      # the C EMD oracle parallelises over clouds
    P = ref.init_params(2020)
    opt = ref.Adam(P)
    g = torch.Generator().manual_seed(2020)
    ex = torch.rand(sample_b, n_half, 3, generator=g) - 0.5
    mi = torch.rand(sample_b, n_half, 3, generator=g) - 0.5
    gt = torch.cat([ex, mi], 1)

    def one():
        pts = torch.stack([ref.generate_points(1, 2 * n_half) for _ in range(sample_b)])   # CPU draws, as the reference
        eps = torch.randn(sample_b, 128)
        ref.train_step(P, opt, ex, mi, gt, pts, eps, emd_coef=emd_coef)
    one()
    t0 = time.perf_counter()
    for _ in range(timed_steps):
        one()
    dt = (time.perf_counter() - t0) / timed_steps
    return {"value": round(sample_b / dt, 3), "unit": "clouds/s", "cores": threads, "kind": "port",
            "sample": f"{timed_steps} timed steps (after 1 warm-up) of the oracle train step (0.05*Chamfer + KLD/B"
                      f"{' + 0.05*EMD/N' if emd_coef else ''}, Adam) at B={sample_b}, "
                      f"existing/missing ({sample_b},{n_half},3), gt ({sample_b},{2 * n_half},3); {dt:.2f} s/step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="clouds per GPU")
    ap.add_argument("--points", type=int, default=2048, help="points per ground-truth cloud")
    ap.add_argument("--no-emd", action="store_true", help="reference-faithful Chamfer-only loss as the headline step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip roofline/breakdown side measurements")
    ap.add_argument("--roofline-only", action="store_true",
                    help="run only the roofline leg (the dominant kernel at the step's shape) and print its object: the "
                         "command profiles/ pairs with `rocprofv3 --kernel-trace --stats`")
    args = ap.parse_args()
    if args.roofline_only:
        torch.cuda.set_device(0)
        print(json.dumps({"roofline": roofline_dominant_kernel(args.batch, args.points // 2)}), flush=True)
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Test hook (not used by the driver): HP_BENCH_BACKEND=gloo HP_BENCH_ONE_DEVICE=1 exercises the multi-rank code
    # path with every rank on cuda:0, so that it can be checked on a 1-GPU box.
    backend = os.environ.get("HP_BENCH_BACKEND", "nccl")
    if os.environ.get("HP_BENCH_ONE_DEVICE"):
        local_rank = 0
    # Test hook (not used by the driver): HP_BENCH_FORCE_EXCHANGE=1 at WORLD_SIZE=1 forms a one-rank RCCL group and runs
    # every collective of the multi-rank step in it (broadcast, bucketed async all-reduce, deferred waits, barrier).
    force_exchange = bool(os.environ.get("HP_BENCH_FORCE_EXCHANGE")) and world == 1
    if force_exchange:
        os.environ.setdefault("MASTER_PORT", "29533")
    if world > 1 or force_exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            # RCCL's stream from the high-priority pool: its own hardware queue (never folded onto the compute stream's),
            # and the few workgroups of a collective are dispatched ahead of the wide GEMMs' instead of queueing behind
            opts = None
            try:
                opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            except Exception:      # older binding: default stream priority
                pass
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank),
                                    pg_options=opts)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    device = torch.device("cuda", local_rank if world > 1 else 0)
    grouped = world > 1 or force_exchange

    from hyperpocket_amd.core.engine import TrainEngine
    from hyperpocket_amd.core.setup import weights_init
    from hyperpocket_amd.model.full_model import FullModel

    n_half = args.points // 2
    torch.manual_seed(2020)                     # settings/config.json.sample:107
    model = FullModel(copy.deepcopy(MODEL_CFG))
    model.apply(weights_init)
    model = model.to(device)
    torch.manual_seed(2020 + rank)              # per-rank streams for eps / decoder points
    emd_coef = 0.0 if args.no_emd else 0.05
    engine = TrainEngine(model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, loss_coef=0.05, emd_coef=emd_coef,
                         force_exchange=force_exchange)
    ex, mi, gt = synth_batch(args.batch, n_half, device, 2020 + rank)

    def sync():
        engine.finish_pending()     # the heads' all-reduce + Adam of the last step (deferred across the step boundary)
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    def run(steps, eng=engine):
        for _ in range(steps):
            out = eng.step(ex, mi, gt, epoch=1)
        return out

    run(args.warmup)
    sync()
    t0 = time.perf_counter()
    out = run(args.steps)
    sync()
    dt = time.perf_counter() - t0
    if grouped:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    loss = out["loss_all"].item()
    assert loss == loss, "NaN loss"
    ms_per_step = dt / args.steps * 1e3
    value = args.batch * world / (ms_per_step * 1e-3)

    if rank == 0:
        line = {
            "metric": "train-step point-clouds/sec at B=64, N=2048 (Chamfer+EMD)" if not args.no_emd
            else "train-step point-clouds/sec at B=64, N=2048 (Chamfer only)",
            "value": round(value, 2), "unit": "clouds/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"HyperPocket 128+128 train step, B={args.batch}/GPU, existing/missing (B,{n_half},3), "
                                   f"gt (B,{args.points},3), loss 0.05*Chamfer + KLD/B" + ("" if args.no_emd else " + 0.05*EMD/N")
                                   + ", Adam lr 1e-4; BASELINE.json configs[1] shape at the metric's B=64",
                       "global_batch": args.batch * world, "points": args.points, "parallelism": f"dp{world}",
                       "params": 43328515},
            "final_loss": loss,
        }
        if not args.no_extras:
            line["roofline"] = roofline_dominant_kernel(args.batch, n_half)
            if world == 1:
                # informational: the reference-faithful Chamfer-only step on the same inputs
                from hyperpocket_amd import ops
                if emd_coef:
                    engine.emd_coef = 0.0
                    run(3)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    run(args.steps)
                    torch.cuda.synchronize()
                    ms2 = (time.perf_counter() - t1) / args.steps * 1e3
                    engine.emd_coef = emd_coef
                    line["breakdown"] = {"chamfer_only_ms_per_step": round(ms2, 4),
                                         "chamfer_only_clouds_per_s": round(args.batch / (ms2 * 1e-3), 2)}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(n_half, emd_coef)
        print(json.dumps(line), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
