"""Data parallelism for the HyperPocket step: one process per GPU, batch sharded across ranks,
parameters replicated, ONE exchange per step — a SUM all-reduce of the flat fp32 gradient buffer
over RCCL/xGMI (the reference has no distributed code at all).

`FlatParameters` re-points every trainable parameter (and, through ops.register_grad_view, every
gradient the HIP backward kernels write) at slices of two contiguous fp32 buffers, laid out in the
order backward produces them: hypernetwork heads (90 % of the bytes) first, then the trunk, then
the encoders.  The head bucket's all-reduce is issued as soon as the hypernetwork backward has
been enqueued, so it travels while the encoder backward (the bulk of the FLOPs) still runs.

Loss semantics under sharding: the Chamfer term is a batch SUM, so gradients are
summed, never averaged; the KLD term is divided by the GLOBAL batch (engine passes B_local*world).
"""
import weakref

import torch
import torch.distributed as dist

from . import ops


HEAD_ROW_MULTIPLE = 8


def _aligned(n, a=4):
    return (n + a - 1) // a * a


class FlatParameters:
    def __init__(self, model):
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        wanted = {id(p) for p in model.parameters()}      # FullModel.parameters() is mode-filtered
        named = [(n, p) for n, p in named if id(p) in wanted]

        def bucket_of(name):
            if name.startswith("hyper_network.output"):
                return 0
            if name.startswith("hyper_network"):
                return 1
            return 2
        # stable sort: module order inside a bucket, except that the heads' weights come first and back to back, then
        # their biases — the five (n_h x 2048) matrices then form ONE (19011 x 2048) matrix and the hypernetwork's
        # head GEMMs (forward, dW, dX) run as single launches (csrc/model.hip: heads_contiguous)
        named.sort(key=lambda np_: (bucket_of(np_[0]), np_[0].endswith(".bias") if bucket_of(np_[0]) == 0 else False))
        self.names = [n for n, _ in named]
        self.params = [p for _, p in named]
        dev = self.params[0].device
        offs, total, bounds = [], 0, {}
        head_w = [(n, p) for n, p in named if bucket_of(n) == 0 and not n.endswith(".bias")]
        # The heads' weights, back to back, are one (rows x cols) matrix; its row count is padded to a multiple of
        # HEAD_ROW_MULTIPLE (zero rows nobody reads) so that 1, 2, 4 or 8 ranks can each own an equal row slice of it
        # (TrainEngine's sharded heads update).  None when the model has no trainable heads / irregular ones.
        self.heads = None
        for i, (n, p) in enumerate(named):
            b = bucket_of(n)
            bounds.setdefault(b, [total, total])
            offs.append(total)
            total = _aligned(total + p.numel())
            if head_w and n == head_w[-1][0]:
                cols = head_w[0][1].size(1)
                rows = sum(q.size(0) for _, q in head_w)
                regular = all(q.dim() == 2 and q.size(1) == cols for _, q in head_w) and cols % 4 == 0 \
                    and total - bounds[0][0] == rows * cols
                if regular:
                    pad_rows = (rows + HEAD_ROW_MULTIPLE - 1) // HEAD_ROW_MULTIPLE * HEAD_ROW_MULTIPLE
                    self.heads = {"lo": bounds[0][0], "rows": rows, "pad_rows": pad_rows, "cols": cols}
                    total = bounds[0][0] + pad_rows * cols
            bounds[b][1] = total
        if self.heads is not None:
            self.heads["hi"] = self.heads["lo"] + self.heads["pad_rows"] * self.heads["cols"]
            self.heads["bias_hi"] = bounds[0][1]          # [hi, bias_hi): the heads' biases
        self.offsets, self.total = offs, total
        self.buckets = [tuple(bounds[b]) for b in sorted(bounds)]
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        keys = []
        with torch.no_grad():
            for p, o in zip(self.params, offs):
                view = self.flat[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view                              # the Parameter object (and any optimizer reference) survives
                keys.append(ops.register_grad_view(p, self.grad[o:o + p.numel()].view(p.shape)))
        # the registry is process-global: drop this owner's entries with the owner (they pin the flat gradient buffer)
        self._finalizer = weakref.finalize(self, ops.drop_grad_views, keys)

    def is_intact(self):
        return all(p.data_ptr() == self.flat.data_ptr() + 4 * o for p, o in zip(self.params, self.offsets))

    def clear_param_grads(self):
        for p in self.params:
            p.grad = None

    def grad_of(self, name):
        i = self.names.index(name)
        p, o = self.params[i], self.offsets[i]
        return self.grad[o:o + p.numel()].view(p.shape)


class GradientReducer:
    """SUM all-reduce of ranges of FlatParameters.grad; `launch` is asynchronous on RCCL's stream (ordered after
    everything enqueued so far on the current stream), `wait` makes the current stream wait for one of them.  Ranges are
    named by a key: a bucket index of FlatParameters.buckets, or any hashable together with explicit bounds."""

    def __init__(self, flat: FlatParameters, process_group=None, force=False):
        self.flat, self.pg = flat, process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        # `force`: issue the collectives even in a one-rank group (a 1-GPU box then exercises the real RCCL calls)
        self.active = self.world > 1 or (force and dist.is_initialized())
        self.works = {}

    def launch(self, key, lo=None, hi=None):
        if not self.active or key in self.works:
            return
        if lo is None:
            lo, hi = self.flat.buckets[key]
        if hi > lo:
            self.works[key] = dist.all_reduce(self.flat.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def all_gather(self, key, out, inp):
        """Asynchronous all_gather_into_tensor under the same bookkeeping (`inp` may be this rank's block of `out`)."""
        if self.active and key not in self.works:
            self.works[key] = dist.all_gather_into_tensor(out, inp, group=self.pg, async_op=True)

    def launch_all(self):
        for b in range(len(self.flat.buckets)):
            self.launch(b)

    def wait(self, key):
        """Make the current stream wait for `key`'s collective (no-op if it was not launched)."""
        w = self.works.pop(key, None)
        if w is not None:
            w.wait()

    def finish(self):
        self.launch_all()
        for k in list(self.works):
            self.wait(k)
