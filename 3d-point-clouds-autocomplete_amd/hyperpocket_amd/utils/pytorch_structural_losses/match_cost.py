"""Drop-in for utils/pytorch_structural_losses/match_cost.py:5-48."""
from torch.autograd import Function

from .StructuralLossesBackend import ApproxMatch, MatchCost, MatchCostGrad


class MatchCostFunction(Function):
    @staticmethod
    def forward(ctx, seta, setb):
        ctx.save_for_backward(seta, setb)
        match, temp = ApproxMatch(seta, setb)
        ctx.match = match           # kept for backward, as the reference does (match_cost.py:20)
        return MatchCost(seta, setb, match)

    @staticmethod
    def backward(ctx, grad_output):
        seta, setb = ctx.saved_tensors
        grada, gradb = MatchCostGrad(seta, setb, ctx.match)
        grad_output_expand = grad_output.unsqueeze(1).unsqueeze(2)
        return grada * grad_output_expand, gradb * grad_output_expand


match_cost = MatchCostFunction.apply
