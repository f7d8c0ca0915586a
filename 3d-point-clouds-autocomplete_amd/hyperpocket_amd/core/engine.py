"""TrainEngine — the build's fast counterpart of one iteration of core/epoch_loops.py:14-39
(zero_grad -> forward -> 0.05*Chamfer + KLD/B [+ EMD] -> backward -> [grad all-reduce] -> Adam),
for callers that own the whole step (bench.py, the DP launcher).  It drives the same FullModel /
ChamferLoss / match_cost objects as the drop-in path; what it adds is the flat parameter/gradient
layout (parallel.py), the overlapped RCCL all-reduce and the fused HIP Adam — and it never
synchronises with the host (the reference does three `.item()` syncs per step, epoch_loops.py:32-36).
"""
import torch
import torch.distributed as dist

from .. import ops
from ..losses.champfer_loss import ChamferLoss
from ..parallel import FlatParameters, GradientReducer
from ..utils.pytorch_structural_losses.match_cost import match_cost


class TrainEngine:
    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, loss_coef=0.05, emd_coef=0.0, process_group=None):
        self.model = model
        self.lr, self.betas, self.eps = lr, betas, eps
        self.loss_coef, self.emd_coef = loss_coef, emd_coef
        self.flat = FlatParameters(model)
        self.reducer = GradientReducer(self.flat, process_group)
        self.world = self.reducer.world
        self.exp_avg = torch.zeros_like(self.flat.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat.flat)
        self.steps = 0
        self._heads_pending = False
        model._pre_hypernet_hook = self.finish_pending     # FullModel.forward calls it right before the hypernetwork
        self.chamfer = ChamferLoss()
        if self.world > 1:
            # replicas start from rank 0's weights
            dist.broadcast(self.flat.flat, src=0, group=process_group)

    def _after_hypernet_backward(self, *_):
        self.reducer.launch(0)
        self.reducer.launch(1)

    def step(self, existing, missing, gt, epoch, points=None, eps_noise=None):
        """One optimisation step on this rank's shard.  Tensors are (B,N,3) device tensors; returns the
        loss terms as 0-dim device tensors (no host sync)."""
        model = self.model
        model.train()
        assert self.flat.is_intact(), "parameters were re-allocated (e.g. .to()) after TrainEngine construction"
        self.flat.clear_param_grads()
        device = gt.device
        # forward() transposes its inputs in place: hand it views it may mutate
        rec, logvar, mu = model(existing.view(existing.shape), None if missing is None else missing.view(missing.shape),
                                list(gt.shape), epoch, device, points=points, eps=eps_noise)
        rec_n3 = rec.permute(0, 2, 1)
        side = None
        if self.emd_coef:
            # Chamfer (VALU-bound, ~0.3 ms) and the EMD sweeps (2 waves/SIMD, VALU pipe ~60 % busy) are independent
            # consumers of the reconstruction: Chamfer goes to a side stream and fills the EMD's idle issue slots
            from ..model.full_model import _side_stream
            cur = torch.cuda.current_stream(device)
            side = _side_stream(model, device)
            side.wait_stream(cur)
            rec.record_stream(side)
            with torch.cuda.stream(side):
                loss_r = self.loss_coef * self.chamfer(gt, rec_n3)
        else:
            loss_r = self.loss_coef * self.chamfer(gt, rec_n3)
        out = {}
        extra = []
        if model.mode.has_generativity():
            kld = ops.kld_loss(logvar, mu, batch=gt.size(0) * self.world)
            extra.append(kld)
            out["loss_kld"] = kld.detach()
        if self.emd_coef:
            emd = self.emd_coef * (match_cost(gt.contiguous(), rec_n3.contiguous()) / float(gt.size(1))).sum()
            extra.append(emd)
            out["loss_emd"] = emd.detach()
            cur.wait_stream(side)
            loss_r.record_stream(cur)
        loss_all = loss_r
        for t in extra:
            loss_all = loss_all + t
        out["loss_r"] = loss_r.detach()
        out["loss_all"] = loss_all.detach()
        if self.world > 1:
            # the hypernetwork's gradients (90 % of the bytes) are complete once its backward has been
            # enqueued; ship them while the encoders' backward still runs
            self._install_overlap_hook()
        loss_all.backward()
        self.steps += 1
        # Exchange + update per bucket.  Trunk and encoders (17 MB) are reduced and updated now; the hypernetwork heads
        # (156 MB, 90 % of the bytes) are only needed again in the NEXT step's hypernetwork forward, which comes after
        # ~1 ms of encoder forward: their all-reduce stays in flight across the step boundary and `finish_pending`
        # (called by FullModel.forward right before the hypernetwork) waits for it and applies their Adam update there.
        self.reducer.launch_all()
        for b in range(1, len(self.flat.buckets)):
            self.reducer.wait(b)
            self._adam(b)
        if self.world > 1:
            self._heads_pending = True
        else:
            self._adam(0)
        return out

    def _adam(self, bucket):
        lo, hi = self.flat.buckets[bucket]
        ops.adam_step(self.flat.flat[lo:hi], self.flat.grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], self.lr,
                      self.betas[0], self.betas[1], self.eps, self.steps)

    def finish_pending(self):
        """Complete the deferred heads update (idempotent).  Call before reading the parameters outside `step`."""
        if self._heads_pending:
            self.reducer.wait(0)
            self._adam(0)
            self._heads_pending = False

    def _install_overlap_hook(self):
        # fires when autograd has finished the HyperNetFunction node, i.e. when the gradient w.r.t. the latent exists
        latent_holder = getattr(self.model, "_last_latent", None)
        if latent_holder is not None and latent_holder.requires_grad:
            latent_holder.register_hook(lambda g: (self._after_hypernet_backward(), g)[1])
