"""``nn_distance(seta, setb) -> (dist1, dist2)``: squared distance from every point of one set to its nearest
neighbour in the other, both directions, differentiable in both sets.

Interface of the reference's wrapper (utils/pytorch_structural_losses/nn_distance.py:6-41): seta (b, n, 3) is the
"dataset" side, setb (b, m, 3) the "query" side; the arg-min indices stay on the autograd context and the backward
is the scatter NNDistanceGrad.  Compute: hp_nndistance / hp_nndistancegrad behind StructuralLossesBackend.
"""
import torch

from . import StructuralLossesBackend as backend


class NNDistanceFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, seta, setb):
        dist1, idx1, dist2, idx2 = backend.NNDistance(seta, setb)
        ctx.save_for_backward(seta, setb)
        ctx.idx1, ctx.idx2 = idx1, idx2          # int32, no gradient: attributes, as upstream keeps them
        return dist1, dist2

    @staticmethod
    def backward(ctx, grad_dist1, grad_dist2):
        seta, setb = ctx.saved_tensors
        return tuple(backend.NNDistanceGrad(seta, setb, ctx.idx1, ctx.idx2,
                                            grad_dist1.contiguous(), grad_dist2.contiguous()))


nn_distance = NNDistanceFunction.apply
