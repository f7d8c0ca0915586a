"""Hypernetwork shell: latent (B, input_size) -> one target-network weight vector per cloud (B, T).

Contract from the reference (model/hyper_network.py:5-43): config keys, state_dict keys ``model.{0,2,4,6,8}.*``
(trunk) and ``output.{0..L}.*`` (one head per target-network layer, each emitting that layer's weight matrix
followed by its bias), construction order = key order so seeded default init matches.  Trunk + heads + the
concatenation are one autograd node (ops.HyperNetFunction).
"""
from torch import nn

from ..ops import HyperNetFunction
from .encoder import _interleave_relu

TRUNK_WIDTHS = (64, 128, 512, 1024, 2048)


def head_sizes(layer_out_channels, with_bias):
    """Length of each target-network layer's slice of the weight vector: (fan_in [+1 for the bias]) * fan_out."""
    chain = [3, *layer_out_channels, 3]
    return [(fan_in + int(with_bias)) * fan_out for fan_in, fan_out in zip(chain[:-1], chain[1:])]


class HyperNetwork(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.input_size = config['input_size']
        self.use_bias = config['use_bias']
        self.relu_slope = config['relu_slope']
        if not self.use_bias:
            raise NotImplementedError("the HIP hypernetwork path expects use_bias=true")

        widths = (self.input_size,) + TRUNK_WIDTHS
        self.model = _interleave_relu([nn.Linear(a, b) for a, b in zip(widths[:-1], widths[1:])], after_last=False)

        sizes = head_sizes(config['target_network_layer_out_channels'], config['target_network_use_bias'])
        heads = [nn.Linear(TRUNK_WIDTHS[-1], s) for s in sizes]
        # frozen heads stay a plain list: unregistered, so absent from parameters() and state_dict(), as upstream
        self.output = heads if config['target_network_freeze_layers_learning'] else nn.ModuleList(heads)

    def forward(self, x):
        trunk = [m for m in self.model if isinstance(m, nn.Linear)]
        heads = list(self.output)
        flat = [m.weight for m in trunk] + [m.bias for m in trunk] + [h.weight for h in heads] + [h.bias for h in heads]
        # (the engine that drives this step may take over the heads' weight gradient: ops.py)
        return HyperNetFunction.apply(x, len(heads), self.__dict__.get("_heads_exchange"), *flat)
