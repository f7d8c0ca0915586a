"""TrainEngine — the build's fast counterpart of one iteration of core/epoch_loops.py:14-39
(zero_grad -> forward -> 0.05*Chamfer + KLD/B [+ EMD] -> backward -> [grad all-reduce] -> Adam),
for callers that own the whole step (bench.py, the DP launcher).  It drives the same FullModel as
the drop-in path and the same loss kernels as ChamferLoss / match_cost (called directly: each yields
its gradient with its value); what it adds is the flat parameter/gradient layout (parallel.py), the
overlapped RCCL all-reduce and the fused HIP Adam — and it never synchronises with the host (the
reference does three `.item()` syncs per step, epoch_loops.py:32-36).
"""
import ctypes
import weakref

import torch
import torch.distributed as dist

from .. import ops
from .._lib import call, check_input, current_stream, load_library
from ..parallel import FlatParameters, GradientReducer


class HeadsShard:
    """Sharded update of the hypernetwork heads under data parallelism (ZeRO-1 over the heads' weight matrix, made cheap
    by the gradient's structure).  The heads are 90 % of the parameters (19011 x 2048 = 156 MB) but their gradient is a
    rank-B product  dW = d theta^T . t5.  Instead of all-reducing 156 MB per step, ranks all-gather the two factors
    (B x (19011 + 2048) floats each: 5.4 MB), rank r forms rows [r*R, (r+1)*R) of the GLOBAL gradient (contraction over
    all world*B clouds: the same flops as its local dW), applies Adam to those rows only (1/world of the heads' Adam
    traffic) and the updated rows are all-gathered in place (78 MB sent per rank at any world size: half the bytes of
    the all-reduce, and nothing of it is on the critical path: it is waited for right before the NEXT step's
    hypernetwork forward).  Every rank must run the same batch size."""

    def __init__(self, flat, reducer, adam_rows):
        h = flat.heads
        self.flat, self.reducer, self.adam_rows = flat, reducer, adam_rows
        self.world, self.rank = reducer.world, reducer.rank
        self.lo, self.rows, self.cols = h["lo"], h["rows"], h["cols"]
        self.R = h["pad_rows"] // self.world                      # rows owned per rank (pad rows: zero weights, zero grads)
        self.r0 = self.rank * self.R
        self.rows_here = max(0, min(self.R, self.rows - self.r0))
        self.hi = h["hi"]
        self._bufs = {}
        self.pending = False
        self.step = 0
        self.weights_in_flight = False

    @staticmethod
    def usable(flat, world):
        return flat.heads is not None and flat.heads["pad_rows"] % world == 0 and flat.heads["cols"] == 2048

    def accepts(self, head_weights):
        base = self.flat.flat.data_ptr() + 4 * self.lo
        off = 0
        for p in head_weights:
            if p.data_ptr() != base + 4 * off:
                return False
            off += p.numel()
        return off == self.rows * self.cols

    def begin(self, grad_theta, t5):
        B = grad_theta.size(0)
        if grad_theta.size(1) != self.rows:
            raise RuntimeError(f"HeadsShard: d theta has {grad_theta.size(1)} columns, the heads have {self.rows} rows")
        key = (B, grad_theta.device)
        if key not in self._bufs:
            f32 = dict(dtype=torch.float32, device=grad_theta.device)
            lib = load_library()
            lib.hp_hypernet_heads_dw_workspace_floats.restype = ctypes.c_long
            self._bufs[key] = (torch.empty((self.world * B, self.rows), **f32), torch.empty((self.world * B, 2048), **f32),
                               torch.empty((lib.hp_hypernet_heads_dw_workspace_floats(),), **f32))
        self._cur = self._bufs[key]
        self._keep = (grad_theta, t5)          # inputs of the in-flight gathers
        self.reducer.all_gather("dtheta", self._cur[0], grad_theta)
        self.reducer.all_gather("t5", self._cur[1], t5)
        self.pending = True

    def update(self):
        """Own rows of the global dW -> Adam on them -> all-gather of the updated rows (asynchronous)."""
        if not self.pending:
            return
        self.reducer.wait("dtheta")
        self.reducer.wait("t5")
        dth, t5, ws = self._cur
        flat = self.flat
        lo = self.lo + self.r0 * self.cols
        hi = lo + self.R * self.cols
        if self.rows_here:
            # this rank's rows of the GLOBAL gradient and their Adam update in one pass (the rows' gradient is never stored;
            # the padding rows past 19011 have zero weights and zero gradients: nothing to do)
            n = self.rows_here * self.cols
            self.adam_rows(dth, t5, self.r0, self.rows_here, lo, lo + n)
        self.reducer.all_gather("heads_w", flat.flat[self.lo:self.hi], flat.flat[lo:hi])
        self.weights_in_flight = True
        self.pending = False
        self._keep = None

    def wait_weights(self):
        if self.weights_in_flight:
            self.reducer.wait("heads_w")
            self.weights_in_flight = False


class FusedHeadsAdam:
    """One GPU: the heads' weight gradient (dW = d theta^T . t5, 156 MB) is never materialised — right behind the
    hypernetwork backward (which still needs the OLD weights for d t5 = d theta . W) one kernel forms it tile by tile in
    registers and applies Adam to W / exp_avg / exp_avg_sq in place (hp_hypernet_heads_dw_adam): 6 x 156 MB of HBM traffic
    instead of the 8 x 156 MB of "write dW, then one Adam pass over it", and the step's last Adam pass shrinks to the
    17 MB of everything else.  Same object protocol as HeadsShard (ops.py: the heads' exchange object)."""

    def __init__(self, engine, own_stream=True):
        h = engine.flat.heads
        self.engine, self.flat = engine, engine.flat
        self.lo, self.hi, self.rows, self.cols = h["lo"], h["hi"], h["rows"], h["cols"]
        # The pass is an HBM stream (186 us at B=64); what follows it in the step — the encoders' backward — is ~70 small
        # dependent launches that leave HBM idle.  On a stream of its own the two overlap; `join` orders it before the next
        # reader of the heads' weights.
        self.stream = torch.cuda.Stream(device=engine.flat.flat.device) if own_stream else None
        self._keep = self._job = None
        self.ran = False           # the pass of the step in flight has been launched (step() then leaves the heads to it)
        self.broken = None         # set by abort(): why the heads are out of step with the rest of the model
        import os
        self.defer = os.environ.get("HP_HEADS_ADAM_DEFER", "1") != "0"
        # Round 4: the pass starts right behind the heads' dX inside the hypernetwork's backward (hp_hypernet_backward_ordered
        # orders `self.stream` there) and occupies only part of the chip (hp_hypernet_heads_dw_adam: persistent workgroups), so
        # that it streams under the trunk's backward and the encoders' tails — launches that leave HBM idle — instead of beside
        # the encoders' gather launch, which is HBM-bound itself.  HP_HEADS_EARLY=0: behind the tails as in round 3.
        self.early = self.stream is not None and os.environ.get("HP_HEADS_EARLY", "1") != "0"

    accepts = HeadsShard.accepts

    def begin(self, grad_theta, t5):
        pass                                   # nothing to exchange

    def finish(self, grad_theta, t5):
        """Called by HyperNetFunction.backward AFTER hp_hypernet_backward has been enqueued (stream order: after d t5).
        With a stream of its own the pass is not launched here: it saturates HBM from every CU for ~180 us, and the
        launches that follow on the compute stream — the encoders' fc/mu/std tails, 372-VGPR latency-built workgroups
        that need empty SIMDs — would sit behind it for that long.  The encoder pair's backward launches it behind its
        tails instead (`launch_ordered`, ops.EncoderPairFunction), beside the matrix-bound conv-stack launches;
        `flush` launches it at the latest when backward is over (modes without the paired encoders)."""
        if self.ran or self._job is not None:
            # The owner's step() (FlatAdam.step / TrainEngine.step) consumes a fused pass and clears `ran`.  A second
            # backward() before that would update the heads' weights AGAIN with the same bias-correction step number —
            # silently (their .grad is None: clip_grad_norm_ / GradScaler never see them).  Fail loudly instead.
            raise RuntimeError(
                "FusedHeadsAdam: backward() reached the hypernetwork heads again before the optimiser's step() consumed the "
                "previous pass (their weights were already updated in place during that backward).  Call step() after every "
                "backward(); for gradient accumulation, skipped steps or a second backward build FlatAdam(model, ..., "
                "fuse_heads=False) / TrainEngine(..., fuse_heads_adam=False).")
        self._job = (grad_theta, t5)
        if self.early:
            self._launch(self.stream)          # (the library has ordered the stream behind the heads' dX already)
        elif self.stream is None or not self.defer:
            self.flush()

    def pending(self):
        return self._job is not None

    def _launch(self, st):
        e = self.engine
        grad_theta, t5 = self._job
        n = self.rows * self.cols
        dev = grad_theta.device
        self._keep, self._job = (grad_theta, t5), None      # alive until join(): the side stream reads them
        with torch.cuda.stream(st):
            if self.early and st is self.stream:      # beside the trunk's backward: the background form (part of the chip)
                call("hp_hypernet_heads_dw_adam_bg", grad_theta.size(0), self.rows, 0, grad_theta, grad_theta.size(1), t5,
                     self.flat.flat[self.lo:self.lo + n], e.exp_avg[self.lo:self.lo + n], e.exp_avg_sq[self.lo:self.lo + n],
                     float(e.lr), float(e.betas[0]), float(e.betas[1]), float(e.eps), int(e._adam_step), 0, current_stream(dev))
            else:
                call("hp_hypernet_heads_dw_adam", grad_theta.size(0), self.rows, 0, grad_theta, grad_theta.size(1), t5,
                     self.flat.flat[self.lo:self.lo + n], e.exp_avg[self.lo:self.lo + n], e.exp_avg_sq[self.lo:self.lo + n],
                     float(e.lr), float(e.betas[0]), float(e.betas[1]), float(e.eps), int(e._adam_step), current_stream(dev))
        self.ran = True

    def launch_ordered(self):
        """The caller has ordered `self.stream` behind the point of the compute stream the pass may start at."""
        if self._job is not None:
            self._launch(self.stream)

    def flush(self):
        if self._job is None:
            return
        cur = torch.cuda.current_stream(self._job[0].device)
        st = self.stream if self.stream is not None else cur
        if st is not cur:
            st.wait_stream(cur)
        self._launch(st)

    def join(self):
        self.flush()
        if self.stream is not None and self._keep is not None:
            torch.cuda.current_stream(self.flat.flat.device).wait_stream(self.stream)
        self._keep = None

    def abort(self):
        """The step in flight failed: a pass that was handed over but not launched is DROPPED (launching it would move
        weights and moments of a step that is not counted; a retry would apply the update twice with one bias-correction
        step number), a pass already running on the side stream is waited for.  With `early` the pass starts inside the
        hypernetwork's backward, i.e. BEFORE the trunk's and the encoders' backward: when those fail, the heads' weights
        and moments have already taken the update of a step nothing else took.  That cannot be undone here (the update is
        in place and the gradient was never stored), so it is recorded: the owner refuses further steps until a
        checkpoint is loaded."""
        self._job = None
        if self.stream is not None and self._keep is not None:
            torch.cuda.current_stream(self.flat.flat.device).wait_stream(self.stream)
        self._keep = None
        if self.ran:
            self.broken = ("the fused dW + Adam pass of the hypernetwork heads had already run (in place) when backward() "
                           "failed: the heads' weights and Adam moments are one update ahead of every other parameter.  "
                           "Reload model and optimiser state from a checkpoint (load_state_dict + load_optimizer_state_dict), "
                           "or build the engine with fuse_heads_adam=False if backward() is expected to fail.")

    def check(self):
        if self.broken:
            raise RuntimeError("engine state inconsistent: " + self.broken)
        if self.ran or self._job is not None:
            raise RuntimeError("engine state inconsistent: a fused heads update from a previous backward() was never "
                               "consumed by a step (did backward() raise?).  Reload a checkpoint.")


class TrainEngine:
    _DEFERRED = (1, 0)   # buckets whose exchange + update cross the step boundary when world > 1: trunk, heads

    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, loss_coef=0.05, emd_coef=0.0, process_group=None,
                 force_exchange=False, shard_heads=True, fuse_heads_adam=True):
        self.model = model
        self.lr, self.betas, self.eps = lr, betas, eps
        self.loss_coef, self.emd_coef = loss_coef, emd_coef
        self.flat = FlatParameters(model)
        self.reducer = GradientReducer(self.flat, process_group, force=force_exchange)
        self.world = self.reducer.world
        self.exchange = self.reducer.active      # world > 1, or a one-rank group asked to run the collectives anyway
        self.exp_avg = torch.zeros_like(self.flat.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat.flat)
        self.steps = 0             # completed optimisation steps
        self._adam_step = 0        # the step number the Adam launches of the step in flight use (bias correction)
        self._heads_pending = False
        self._deferred_losses = None
        import os as _os
        self._join_early = _os.environ.get("HP_HEADS_JOIN_AT_STEP_END", "0") == "1"
        self._loss_side_stream = _os.environ.get("HP_LOSS_SIDE_STREAM", "1") != "0"
        self._predraw_on = _os.environ.get("HP_PREDRAW", "1") != "0"
        self._next_eps = self._next_points = None      # pre-drawn for the next step (step / _predraw)
        self._draws_eps = self._draws_points = False   # the caller has left a draw to the engine at least once
        # The model only holds WEAK references to its engine (a dropped engine must not stay pinned — with its four flat
        # 173 MB buffers — by the model's hooks), and a new engine on the same model takes the hooks over.
        prev = model.__dict__.get("_engine_ref")
        prev = prev() if prev is not None else None
        if prev is not None:
            prev.close()
        me = weakref.ref(self)
        model._engine_ref = me

        def pre_hypernet():                                # FullModel.forward calls it right before the hypernetwork
            e = me()
            if e is not None:
                e.finish_pending()

        def pre_state_dict(*_):
            # A reader of the parameters outside `step` (model.state_dict(), torch.save: core/main.py:164) must not see the
            # hypernetwork one step behind the encoders or half-gathered rows: flush the deferred updates first.
            e = me()
            if e is not None:
                e.synchronize()
        model._pre_hypernet_hook = pre_hypernet
        self._sd_hook = model.register_state_dict_pre_hook(pre_state_dict)
        self._consts = {}
        # the heads' update: sharded over the ranks (HeadsShard) when the layout allows it, else all-reduced like the rest
        self.shard = None
        if self.exchange and shard_heads and HeadsShard.usable(self.flat, self.world):
            self.shard = HeadsShard(self.flat, self.reducer, self._adam_heads_rows)
        # one rank: the heads' dW and its Adam update are one kernel (the gradient of the heads' weights is then never
        # stored: their .grad stays None)
        self.fused = None
        if not self.exchange and fuse_heads_adam and HeadsShard.usable(self.flat, 1):
            import os
            self.fused = FusedHeadsAdam(self, own_stream=os.environ.get("HP_HEADS_ADAM_STREAM", "1") != "0")
        if self.exchange:
            # replicas start from rank 0's weights
            dist.broadcast(self.flat.flat, src=0, group=process_group)

    def close(self):
        """Detach from the model: remove the state_dict hook and the pre-hypernetwork hook (idempotent).  Called when another
        engine attaches to the same model; the flat parameter layout stays (the parameters keep pointing at it)."""
        self.finish_pending()
        if self._sd_hook is not None:
            self._sd_hook.remove()
            self._sd_hook = None
        ref = self.model.__dict__.get("_engine_ref")
        if ref is not None and ref() is self:
            self.model._pre_hypernet_hook = None
            self.model._engine_ref = None

    def _after_hypernet_backward(self, *_):
        if self.shard is not None:
            # d theta / t5 were gathered under the hypernetwork's own backward launches; the heads' rows are updated
            # and on their way before the encoders' backward is even enqueued
            self.shard.update()
            lo, hi = self.shard.hi, self.flat.buckets[1][1]      # heads' biases + trunk: plain all-reduce, deferred
            self.reducer.launch("small", lo, hi)
            return
        self.reducer.launch(0)
        self.reducer.launch(1)

    def step(self, existing, missing, gt, epoch, points=None, eps_noise=None):
        """One optimisation step on this rank's shard.  Tensors are (B,N,3) device tensors; returns the
        loss terms as 0-dim device tensors (no host sync)."""
        model = self.model
        if not model.training:
            model.train()          # (recursive over ~50 modules: 0.2 ms of host time when called every step)
        assert self.flat.is_intact(), "parameters were re-allocated (e.g. .to()) after TrainEngine construction"
        self.flat.clear_param_grads()
        device = gt.device
        # who takes the heads' weight gradient of THIS step's hypernetwork node (state of this model, not of the process)
        if self.fused is not None:
            # (a pass left over from a failed step must not be cleared silently — and it raises BEFORE the hooks below are
            #  installed on the model: a raise behind them would leave a later backward outside the engine routed into this object)
            self.fused.check()
        exch = self.shard if (self.exchange and self.shard is not None) else (self.fused if not self.exchange else None)
        model.hyper_network._heads_exchange = exch
        model._after_encoder_tails = self.fused if (self.fused is not None and self.fused.stream is not None) else None
        # The step's own random draws — eps of the VAE encoder, the decoder's input points — depend on nothing: the PREVIOUS step
        # drew them on the side stream, beside its EMD sweeps (_predraw), so that the two small launches (4 + 8 us with their
        # gaps) are not on the compute stream between the optimiser and the first conv launch / in front of the decoder.  The
        # n-th draw of each sequence is the n-th value either way; draws a caller injects (tests) leave the pre-drawn ones for
        # the next step that does not.
        want = (gt.size(0), gt.size(1), epoch)
        self._draws_eps, self._draws_points = self._draws_eps or eps_noise is None, self._draws_points or points is None
        if self._next_eps is not None and self._next_eps[0] != want[0]:
            self._next_eps = None              # drawn for another batch size: drop it (it would block every later pre-draw)
        if self._next_points is not None and self._next_points[0] != want:
            self._next_points = None           # ... another shape or epoch (the hollow's radius follows the epoch)
        if eps_noise is None and self._next_eps is not None:
            eps_noise, self._next_eps = self._next_eps[1], None
        if points is None and self._next_points is not None:
            points, self._next_points = self._next_points[1], None
        self._predraw_for = (want, eps_noise is not None, points is not None)
        # forward() transposes its inputs in place: hand it views it may mutate
        rec, logvar, mu = model(existing.view(existing.shape), None if missing is None else missing.view(missing.shape),
                                list(gt.shape), epoch, device, points=points, eps=eps_noise)
        rec_n3 = rec.permute(0, 2, 1)
        roots, root_grads, out = self._losses_and_gradients(gt, rec_n3, logvar, mu)
        self._adam_step = self.steps + 1      # `steps` itself moves only once backward has succeeded
        if self.exchange:
            # the hypernetwork's gradients (90 % of the bytes) are complete once its backward has been
            # enqueued; ship them while the encoders' backward still runs
            self._install_overlap_hook()
        try:
            torch.autograd.backward(roots, root_grads)
        except BaseException:
            if self.fused is not None:
                self.fused.abort()              # drop an unlaunched pass; the side stream must not outlive the failed step
            raise
        else:
            if self.fused is not None:
                self.fused.flush()              # (modes without the paired encoders' backward; a no-op otherwise)
        finally:
            model.hyper_network._heads_exchange = None
            model._after_encoder_tails = None
            args, self._deferred_losses = self._deferred_losses, None
            if args is not None:
                call("hp_step_losses", *args, current_stream(device))
        self.steps = self._adam_step
        # Exchange + update per bucket.  The encoders' bucket (6.6 MB) is reduced and updated now: the next step starts
        # with it.  The hypernetwork's buckets (heads 156 MB, trunk 11 MB: 96 % of the bytes) are only needed again in the
        # NEXT step's hypernetwork forward, which comes after ~1 ms of encoder forward: their all-reduces stay in flight
        # across the step boundary and `finish_pending` (called by FullModel.forward right before the hypernetwork)
        # waits for them and applies their Adam updates there.
        if not self.exchange:
            # nothing to exchange: one pass over the flat buffer (minus the heads' weights when their update was fused
            # into the hypernetwork backward)
            # (the heads are skipped only if their fused pass really ran for this step)
            self._adam_range(self.fused.hi if (self.fused is not None and self.fused.ran) else 0, self.flat.total)
            if self.fused is not None:
                self.fused.ran = False          # consumed: this step is counted for the heads and for everything else
            # The heads' pass (936 MB of HBM traffic whatever the batch: ~180 us) is joined where its result is next READ —
            # `finish_pending`, right before the next step's hypernetwork forward — not here: at B = 64 it ends inside the
            # encoders' backward anyway, at B = 32 the compute stream would otherwise idle ~130 us for it at the step end
            # instead of starting the next step's encoder forward.  HP_HEADS_JOIN_AT_STEP_END=1 restores the early join.
            if self.fused is not None and self._join_early:
                self.fused.join()
            self._heads_pending = False
            return out
        if self.shard is not None:
            self._after_hypernet_backward()       # no-op when the latent's hook already ran it
            self.reducer.launch(2)
            self.reducer.wait(2)
            self._adam(2)
            self._heads_pending = True
            return out
        self.reducer.launch_all()
        for b in range(len(self.flat.buckets) - 1, -1, -1):
            if b not in self._DEFERRED:
                self.reducer.wait(b)
                self._adam(b)
        self._heads_pending = True
        return out

    def _losses_and_gradients(self, gt, rec_n3, logvar, mu):
        """loss_all = loss_coef*Chamfer(gt, rec) + KLD/B [+ emd_coef*sum_b EMD_b/N] (core/epoch_loops.py:26-31) and its
        gradients w.r.t. the model outputs, straight from the fused kernels: every loss kernel here produces its
        gradient alongside its value, so the autograd graph starts at the model outputs (no loss nodes, no scalar glue
        kernels between the loss and the first backward GEMM).  Returns (roots, their gradients, the loss terms)."""
        model = self.model
        dev = gt.device
        B, N = gt.size(0), gt.size(1)
        f32 = dict(dtype=torch.float32, device=dev)
        lib = load_library()
        gt_c, rec_c = gt.contiguous(), rec_n3.contiguous()
        check_input(gt_c, "gt")
        check_input(rec_c, "reconstruction")
        if gt_c.shape != rec_c.shape or gt_c.size(2) != 3:
            raise RuntimeError(f"TrainEngine: reconstruction {tuple(rec_c.shape)} vs gt {tuple(gt_c.shape)}")
        has_kld = model.mode.has_generativity()
        if dev not in self._consts:
            self._consts[dev] = (torch.full((), float(self.loss_coef), **f32), torch.ones((), **f32))
        c_cd, one = self._consts[dev]
        cur = torch.cuda.current_stream(dev)
        # Everything that outlives the side-stream section is allocated here, on the compute stream: the join below orders
        # the two streams, so no tensor needs record_stream() — each of those costs a hipEventRecord on the compute
        # queue when the block is freed, i.e. a ~5 us bubble between two kernels.
        cd = torch.empty((), **f32)
        g_rec = torch.empty_like(rec_c)
        kld = g_lv = g_mu = lv_c = mu_c = None
        if has_kld:
            lv_c, mu_c = logvar.contiguous(), mu.contiguous()
            kld = torch.empty((), **f32)
            g_lv, g_mu = torch.empty_like(lv_c), torch.empty_like(mu_c)
        side = None
        if self.emd_coef and self._loss_side_stream:
            # Chamfer / KLD (VALU-bound, ~0.15 ms) and the EMD sweeps (2 waves/SIMD, VALU pipe ~60 % busy) are
            # independent consumers of the model outputs: the small ones go to a side stream and fill the EMD's idle
            # issue slots
            from ..model.full_model import _side_stream
            side = _side_stream(model, dev)
            side.wait_stream(cur)
        with torch.cuda.stream(side if side is not None else cur):
            st = current_stream(dev)
            dist1, dist2 = torch.empty((B, N), **f32), torch.empty((B, N), **f32)      # temporaries of this section
            idx1 = torch.empty((B, N), dtype=torch.int32, device=dev)
            idx2 = torch.empty((B, N), dtype=torch.int32, device=dev)
            part = torch.empty((lib.hp_chamfer_workspace_floats(B, N, N),), **f32)
            # ChamferLoss()(gt, reconstruction): losses/champfer_loss.py:11-17 with preds = gt (core/epoch_loops.py:26)
            call("hp_chamfer_forward", B, N, gt_c, N, rec_c, dist1, idx1, dist2, idx2, part, cd, st)
            call("hp_chamfer_backward", B, N, gt_c, N, rec_c, idx1, idx2, c_cd, None, g_rec, st)
            if has_kld:
                n_el, batch = ctypes.c_long(mu_c.numel()), B * self.world
                call("hp_kld_forward", n_el, batch, lv_c, mu_c, kld, st)
                call("hp_kld_backward", n_el, batch, lv_c, mu_c, one, g_lv, g_mu, st)
            del dist1, dist2, idx1, idx2, part
            if side is not None and self._predraw_on:
                self._predraw(dev)      # next step's eps and decoder points, on this stream (no dependency, nothing waits for them)
        cost = None
        c_emd = 0.0
        if self.emd_coef:
            c_emd = float(self.emd_coef) / float(N)
            lib.hp_emd_partials_floats.restype = ctypes.c_long
            temp = torch.empty((B, 4 * N), **f32)
            ws = torch.empty((max(1, lib.hp_approxmatch_workspace_floats(B, N, N)),), **f32)
            epart = torch.empty((max(1, lib.hp_emd_partials_floats(B, N, N)),), **f32)
            cost = torch.empty((B,), **f32)
            # match_cost(gt, reconstruction): cost and d cost / d reconstruction from the same sweeps; the gradient sweep adds
            # c_emd * (its term) onto the Chamfer gradient the side stream left in g_rec, behind an event on that stream
            # (which also joins the KLD gradients): no axpy launch, no separate stream join
            call("hp_emd_forward_acc", B, N, N, gt_c, rec_c, temp, ws, epart, cost, g_rec, float(c_emd), current_stream(dev),
                 ctypes.c_void_p(side.cuda_stream) if side is not None else None)
        terms = torch.empty((4,), **f32)
        # the scalar loss terms are nobody's input: their launch is deferred behind the backward's launches (step()), so it
        # does not sit between the EMD and the first backward kernel
        self._deferred_losses = (B, cd, kld, cost, float(self.loss_coef), c_emd, terms)
        out = {"loss_r": terms[0], "loss_all": terms[3]}
        if has_kld:
            out["loss_kld"] = terms[1]
        if self.emd_coef:
            out["loss_emd"] = terms[2]
        roots, grads = [rec_n3], [g_rec]
        if has_kld:
            roots += [logvar, mu]
            grads += [g_lv.view_as(logvar), g_mu.view_as(mu)]
        return roots, grads, out

    def discard_predrawn(self):
        """Drop the draws made ahead for the next step.  The pre-draw advances torch's device generator and the point sampler's
        counter ONE step early: an RNG state saved between two steps is one draw ahead of the parameters saved with it, so a run
        resumed from such a checkpoint re-draws what was already drawn unless the saver calls this first and the resumed run
        starts with HP_PREDRAW=0 semantics for its first step — or, simpler, saves and restores RNG state right after
        construction / this call.  (Bit-identical resume needs the same pre-draw setting in both runs.)"""
        self._next_eps = self._next_points = None

    def _predraw(self, dev):
        """Next step's random draws (called on the side stream).  Only what this step did NOT get injected, and only for the
        shapes of this step (another batch size or epoch next time: the pre-drawn tensors are ignored and drawn again then —
        the device sampler's counter has moved, which shifts its sequence but not its law).
        The two tensors are the one exception to "what outlives the side-stream section is allocated on the compute stream"
        (_losses_and_gradients): they are allocated on the side stream, consumed on the compute stream in the next step (behind
        the join that ends this section) and freed there; their blocks return to the SIDE stream's pool, whose every later user
        first does side.wait_stream(compute) — that wait is what makes the reuse safe without record_stream()."""
        model = self.model
        (B, N, epoch), eps_given, pts_given = self._predraw_for
        auto_eps = getattr(self, "_auto_eps", None)
        if auto_eps is None:      # does this model draw eps itself in training? (HyperPocket / HyperCloud: the VAE encoder)
            auto_eps = self._auto_eps = bool(model.mode.vae_input)
        if auto_eps and self._next_eps is None and (self._draws_eps or not eps_given):
            enc = model.random_encoder
            self._next_eps = (B, torch.randn((B, enc.output_size), dtype=torch.float32, device=dev))
        if model.point_sampler == "device" and self._next_points is None and (self._draws_points or not pts_given):
            self._next_points = ((B, N, epoch), model._draw_points(epoch, B, N, dev))

    def _adam(self, bucket):
        self._adam_range(*self.flat.buckets[bucket])

    def _adam_range(self, lo, hi):
        ops.adam_step(self.flat.flat[lo:hi], self.flat.grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], self.lr,
                      self.betas[0], self.betas[1], self.eps, self._adam_step)

    def _adam_heads_rows(self, dtheta_all, t5_all, r0, rows, lo, hi):
        call("hp_hypernet_heads_dw_adam", dtheta_all.size(0), rows, r0, dtheta_all, dtheta_all.size(1), t5_all,
             self.flat.flat[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], float(self.lr), float(self.betas[0]),
             float(self.betas[1]), float(self.eps), int(self._adam_step), current_stream(dtheta_all.device))

    def finish_pending(self):
        """Complete the deferred hypernetwork updates (idempotent).  Call before reading the parameters outside `step`."""
        if self.fused is not None:
            self.fused.join()          # (one GPU: the fused dW + Adam pass of the heads, if still in flight)
        if self._heads_pending and self.shard is not None:
            self.reducer.wait("small")
            self._adam_range(self.shard.hi, self.flat.buckets[1][1])
            self.shard.wait_weights()
            self._heads_pending = False
        if self._heads_pending:
            for b in self._DEFERRED:
                if b < len(self.flat.buckets):
                    self.reducer.wait(b)
                    self._adam(b)
            self._heads_pending = False

    def synchronize(self):
        """finish_pending + wait for the device: after it the parameters are the state after `steps` whole steps."""
        self.finish_pending()
        if self.flat.flat.is_cuda:       # (CPU tensors: the gloo host-logic tests; the collectives' waits above are blocking there)
            torch.cuda.current_stream(self.flat.flat.device).synchronize()

    # ------------------------------------------------------------------ optimiser checkpoints
    def _moment_views(self, buf):
        """Per-parameter views of a flat moment buffer, in `model.parameters()` order — the order the reference builds its
        Adam in (core/main.py:62-66), hence the index space of its `{epoch}_O.pth` files."""
        off = {id(p): o for p, o in zip(self.flat.params, self.flat.offsets)}
        return [buf[off[id(p)]:off[id(p)] + p.numel()].view(p.shape) for p in self.model.parameters()]

    def _full_moments(self):
        """exp_avg / exp_avg_sq with every rank's rows of the sharded heads gathered (collective under DP)."""
        m, v = self.exp_avg.clone(), self.exp_avg_sq.clone()
        if self.shard is not None and self.world > 1:
            sh = self.shard
            lo = sh.lo + sh.r0 * sh.cols
            hi = lo + sh.R * sh.cols
            for full, own in ((m, self.exp_avg), (v, self.exp_avg_sq)):
                dist.all_gather_into_tensor(full[sh.lo:sh.hi], own[lo:hi].clone(), group=self.reducer.pg)
        return m, v

    def optimizer_state_dict(self):
        """The optimiser state in `torch.optim.Adam.state_dict()` format, i.e. what the reference saves as `{epoch}_O.pth`
        (core/main.py:165) and restores with `optimizer.load_state_dict` (core/setup.py:96-97): loadable into a
        torch.optim.Adam over `full_model.parameters()` and back.  Under data parallelism the heads' moments live
        row-sharded on the ranks: every rank must call this (one all-gather), every rank gets the full state."""
        self.synchronize()
        m, v = self._full_moments()
        params = list(self.model.parameters())
        group = dict(torch.optim.Adam([torch.zeros(1)], lr=self.lr, betas=self.betas, eps=self.eps).state_dict()["param_groups"][0])
        group["params"] = list(range(len(params)))
        state = {}
        if self.steps > 0:
            for i, (mi, vi) in enumerate(zip(self._moment_views(m), self._moment_views(v))):
                state[i] = {"step": torch.tensor(float(self.steps)), "exp_avg": mi.clone(), "exp_avg_sq": vi.clone()}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd):
        """Inverse of optimizer_state_dict (also accepts a reference `{epoch}_O.pth`).  Parameters without an entry (never
        stepped: real_encoder.std_layer in HyperPocket mode) keep zero moments."""
        self.synchronize()
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps = float(g["lr"]), tuple(float(b) for b in g["betas"]), float(g["eps"])
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        steps = 0
        mv, vv = self._moment_views(self.exp_avg), self._moment_views(self.exp_avg_sq)
        for i, st in sd["state"].items():
            i = int(i)
            mv[i].copy_(st["exp_avg"].to(mv[i].device).view_as(mv[i]))
            vv[i].copy_(st["exp_avg_sq"].to(vv[i].device).view_as(vv[i]))
            steps = max(steps, int(float(st["step"])))
        self.steps = self._adam_step = steps
        if self.fused is not None:       # a restored checkpoint is a consistent state again
            self.fused.broken, self.fused.ran, self.fused._job = None, False, None

    def _install_overlap_hook(self):
        # fires when autograd has finished the HyperNetFunction node, i.e. when the gradient w.r.t. the latent exists
        latent_holder = getattr(self.model, "_last_latent", None)
        if latent_holder is not None and latent_holder.requires_grad:
            latent_holder.register_hook(lambda g: (self._after_hypernet_backward(), g)[1])
