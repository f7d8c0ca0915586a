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
import torch
import torch.distributed as dist

from . import ops


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
        for n, p in named:
            b = bucket_of(n)
            bounds.setdefault(b, [total, total])
            offs.append(total)
            total = _aligned(total + p.numel())
            bounds[b][1] = total
        self.offsets, self.total = offs, total
        self.buckets = [tuple(bounds[b]) for b in sorted(bounds)]
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(self.params, offs):
                view = self.flat[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view                              # the Parameter object (and any optimizer reference) survives
                ops.register_grad_view(p, self.grad[o:o + p.numel()].view(p.shape))

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
    """SUM all-reduce of FlatParameters.grad in buckets; `launch(b)` is asynchronous on RCCL's
    stream (ordered after everything enqueued so far on the compute stream), `finish()` makes the
    compute stream wait for all of them."""

    def __init__(self, flat: FlatParameters, process_group=None, force=False):
        self.flat, self.pg = flat, process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # `force`: issue the collectives even in a one-rank group (a 1-GPU box then exercises the real RCCL calls)
        self.active = self.world > 1 or (force and dist.is_initialized())
        self.works = {}
        self.launched = set()

    def launch(self, bucket):
        if not self.active or bucket in self.launched:
            return
        lo, hi = self.flat.buckets[bucket]
        self.launched.add(bucket)
        self.works[bucket] = dist.all_reduce(self.flat.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def launch_all(self):
        for b in range(len(self.flat.buckets)):
            self.launch(b)

    def wait(self, bucket):
        """Make the current stream wait for `bucket`'s all-reduce (no-op if it was not launched / world == 1)."""
        if bucket in self.launched:
            self.works.pop(bucket).wait()
            self.launched.discard(bucket)

    def finish(self):
        self.launch_all()
        for b in list(self.launched):
            self.wait(b)
