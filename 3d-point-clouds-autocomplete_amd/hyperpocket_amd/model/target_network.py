"""Drop-in for model/target_network.py:6-45.

The reference builds one ``TargetNetwork`` nn.Module per cloud inside a Python loop
(model/full_model.py:70-74).  ``TargetNetwork`` keeps that constructor and single-cloud ``forward``
(it runs the batched kernel with B=1); FullModel calls ``target_network_batched`` once for all clouds.
"""
import torch
import torch.nn as nn

from ..ops import TargetNetworkFunction


def target_network_batched(config, weights, points):
    """weights (B, T), points (B, N, 3) -> (B, N, 3); T must equal the layout's length
    (model/target_network.py:29)."""
    if not config['use_bias']:
        raise NotImplementedError("the HIP target-network path expects use_bias=true")
    return TargetNetworkFunction.apply(weights, points, tuple(config['layer_out_channels']))


class TargetNetwork(nn.Module):
    def __init__(self, config, weights):
        super().__init__()
        self.use_bias = config['use_bias']
        self.config = config
        self.weights = weights
        out_ch = config['layer_out_channels']
        dims = [3] + list(out_ch) + [3]
        total = sum(dims[i] * dims[i + 1] + (dims[i + 1] if self.use_bias else 0) for i in range(len(dims) - 1))
        assert total == len(weights)

    def forward(self, x):
        return target_network_batched(self.config, self.weights.unsqueeze(0), x.unsqueeze(0))[0]
