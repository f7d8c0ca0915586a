"""FlatAdam — a `torch.optim.Optimizer` for the drop-in route.

The reference builds `torch.optim.Adam(full_model.parameters(), lr=1e-4)` (core/main.py:62-66) and its loop calls
`optimizer.zero_grad()` / `loss.backward()` / `optimizer.step()` (core/epoch_loops.py:15-39).  With the drop-in modules that
route works unchanged, but the optimiser side then costs what it costs in PyTorch: ~100 parameter tensors stepped by the
foreach kernels, the 156 MB gradient of the hypernetwork heads written by backward and read again by the step.

`FlatAdam(full_model, lr=...)` is the one-line replacement for that constructor: the same class
protocol (param_groups for `StepLR`, `state_dict()` in torch.optim.Adam's own format, `load_state_dict` of the
reference's `{epoch}_O.pth`), and behind it the engine's machinery:

* the parameters are re-pointed at ONE flat fp32 buffer (`parallel.FlatParameters`; names, shapes and the `state_dict`
  of the model are untouched) and the HIP backward kernels write their gradients straight into its twin, so `step()` is
  one `hp_adam_step` launch over the buffer;
* with `fuse_heads=True` (default) the gradient of the hypernetwork heads' weights (19011 x 2048) is never stored:
  one kernel behind the hypernetwork's backward forms it tile by tile and applies Adam in place
  (`hp_hypernet_heads_dw_adam`, `core.engine.FusedHeadsAdam`).  The heads' `.grad` then stays `None` and their weights
  move during `backward()` — equivalent for every loop that calls `step()` after each `backward()` (the reference's
  does).  A loop that does NOT — a skipped step after a non-finite loss, gradient accumulation, a second backward —
  would update the heads twice with one bias-correction step number; that is detected and raised (at the second
  backward, or at the next `zero_grad()`), never silent.  Pass `fuse_heads=False` for such loops or for gradient
  inspection (`clip_grad_norm_` / `GradScaler` cannot see the heads while their `.grad` is None).

Update arithmetic: `hp_adam_step` performs torch.optim.Adam's operations in its order (wd = 0, amsgrad = False — the
reference's settings); tests/test_model_gpu.py compares the two over several steps.
"""
import torch

from . import ops
from .core.engine import FusedHeadsAdam, HeadsShard
from .parallel import FlatParameters


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, fuse_heads=True):
        if weight_decay != 0 or amsgrad:
            raise ValueError("FlatAdam implements the reference's setting: weight_decay=0, amsgrad=False")
        if not hasattr(model, "named_parameters"):
            raise TypeError("FlatAdam takes the model (it re-points its parameters at one flat buffer), not a parameter list")
        self.model = model
        self.flat = FlatParameters(model)
        super().__init__(list(model.parameters()), dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False))
        self.exp_avg = torch.zeros_like(self.flat.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat.flat)
        self._param_ptrs = [self.flat.flat.data_ptr() + 4 * o for o in self.flat.offsets]
        self._grad_ptrs = [self.flat.grad.data_ptr() + 4 * o for o in self.flat.offsets]
        self.steps = 0
        self._adam_step = 1        # the step number the launches of the iteration in flight use (bias correction)
        self.fused = None
        if fuse_heads and HeadsShard.usable(self.flat, 1) and hasattr(model, "hyper_network"):
            self.fused = FusedHeadsAdam(self, own_stream=True)
            # the hypernetwork's autograd node hands the heads' weight gradient to `fused` (ops.HyperNetFunction) and the
            # paired encoders' backward launches its pass behind their tails (ops.EncoderPairFunction)
            model.hyper_network._heads_exchange = self.fused
            model._after_encoder_tails = self.fused

    # FusedHeadsAdam reads these off its owner
    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    @property
    def betas(self):
        return self.param_groups[0]["betas"]

    @property
    def eps(self):
        return self.param_groups[0]["eps"]

    def zero_grad(self, set_to_none=True):
        if self.fused is not None and (self.fused.ran or self.fused.pending()):
            raise RuntimeError(
                "FlatAdam.zero_grad(): the previous backward() already applied (or queued) the fused Adam update of the "
                "hypernetwork heads, but step() was not called for it — the rest of the model would miss that step.  Call "
                "step() after every backward(), or build FlatAdam(model, ..., fuse_heads=False).  (If that backward() "
                "raised part-way, the heads are one update ahead of the other parameters: reload model and optimiser "
                "state from a checkpoint; load_state_dict() on this optimiser clears the condition.)")
        self.flat.clear_param_grads()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        flat = self.flat
        # (the per-parameter checks below run every iteration of the caller's loop: plain integer compares against the
        #  addresses computed once in __init__, no tensor views unless something has to be copied or zeroed)
        for p, want in zip(flat.params, self._param_ptrs):
            if p.data_ptr() != want:
                raise RuntimeError("FlatAdam: parameters were re-allocated (e.g. model.to()) after the optimiser was built")
        # a gradient produced outside the HIP autograd nodes (or accumulated into a fresh tensor) is copied into its slot;
        # a parameter without gradient contributes zeros (torch.optim.Adam skips it: with zero moments the update is zero)
        fused = self.fused
        for p, o, want in zip(flat.params, flat.offsets, self._grad_ptrs):
            g = p.grad
            if g is None:
                if not (fused is not None and fused.lo <= o < fused.hi and fused.ran):
                    flat.grad[o:o + p.numel()].zero_()
            elif g.data_ptr() != want:
                flat.grad[o:o + p.numel()].copy_(g.reshape(-1))
        lo = 0
        if self.fused is not None:
            self.fused.flush()          # (a backward without the paired encoders' node: launch the pass now)
            if self.fused.ran:
                lo = self.fused.hi
        b1, b2 = self.betas
        ops.adam_step(flat.flat[lo:flat.total], flat.grad[lo:flat.total], self.exp_avg[lo:flat.total],
                      self.exp_avg_sq[lo:flat.total], self.lr, b1, b2, self.eps, self._adam_step)
        if self.fused is not None:
            self.fused.join()
            self.fused.ran = False
        self.steps = self._adam_step
        self._adam_step = self.steps + 1
        return loss

    # ---- checkpoints in torch.optim.Adam's own format (the reference's {epoch}_O.pth, core/main.py:165)
    def _moment_views(self, buf):
        off = {id(p): o for p, o in zip(self.flat.params, self.flat.offsets)}
        return [buf[off[id(p)]:off[id(p)] + p.numel()].view(p.shape) for p in self.model.parameters()]

    def state_dict(self):
        torch.cuda.current_stream(self.flat.flat.device).synchronize()
        group = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        group["params"] = list(range(len(self.param_groups[0]["params"])))
        state = {}
        if self.steps > 0:
            for i, (m, v) in enumerate(zip(self._moment_views(self.exp_avg), self._moment_views(self.exp_avg_sq))):
                state[i] = {"step": torch.tensor(float(self.steps)), "exp_avg": m.clone(), "exp_avg_sq": v.clone()}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        g = sd["param_groups"][0]
        for k in ("lr", "betas", "eps"):
            self.param_groups[0][k] = g[k] if k != "betas" else tuple(g[k])
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        steps = 0
        mv, vv = self._moment_views(self.exp_avg), self._moment_views(self.exp_avg_sq)
        for i, st in sd["state"].items():
            i = int(i)
            mv[i].copy_(st["exp_avg"].to(mv[i].device).view_as(mv[i]))
            vv[i].copy_(st["exp_avg_sq"].to(vv[i].device).view_as(vv[i]))
            steps = max(steps, int(float(st["step"])))
        self.steps, self._adam_step = steps, steps + 1
        if self.fused is not None:       # a restored checkpoint is a consistent state again
            self.fused.abort()
            self.fused.broken, self.fused.ran = None, False
