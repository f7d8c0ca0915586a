"""Drop-in for utils/pytorch_structural_losses/nn_distance.py:6-41."""
from torch.autograd import Function

from .StructuralLossesBackend import NNDistance, NNDistanceGrad


class NNDistanceFunction(Function):
    @staticmethod
    def forward(ctx, seta, setb):
        # set1 : batch_size * #dataset_points * 3 ; set2 : batch_size * #query_points * 3
        ctx.save_for_backward(seta, setb)
        dist1, idx1, dist2, idx2 = NNDistance(seta, setb)
        ctx.idx1 = idx1
        ctx.idx2 = idx2
        return dist1, dist2

    @staticmethod
    def backward(ctx, grad_dist1, grad_dist2):
        seta, setb = ctx.saved_tensors
        grada, gradb = NNDistanceGrad(seta, setb, ctx.idx1, ctx.idx2, grad_dist1.contiguous(), grad_dist2.contiguous())
        return grada, gradb


nn_distance = NNDistanceFunction.apply
