"""PointNet-style encoder shell over hp_encoder_forward / hp_encoder_backward.

Contract taken from the reference (model/encoder.py:5-53): constructor ``Encoder(config, is_vae)``, the
state_dict key set ``conv.{0,2,4,6,8}.{weight,bias}``, ``fc.0.*``, ``mu_layer.*``, ``std_layer.*`` with
Conv1d-shaped (Cout, Cin, 1) weights, and the return convention (VAE: ``(z, mu, exp(logvar))``; plain:
``mu``).  Modules are created in key order so a seeded construction draws the same default-init numbers.
This class only owns parameters; the arithmetic is one autograd node (ops.EncoderFunction).
"""
import torch
from torch import nn

from ..ops import EncoderFunction

# pointwise stack: channel widths of the 1x1 convolutions; a ReLU follows every one but the last
POINT_WIDTHS = (3, 64, 128, 256, 512, 512)
POOLED = POINT_WIDTHS[-1]


def _interleave_relu(layers, after_last):
    """[l0, relu, l1, relu, ...] — the ReLU slots carry no parameters; they exist so that the layers sit at the even
    indices the checkpoints name."""
    seq = []
    for i, layer in enumerate(layers):
        seq.append(layer)
        if after_last or i + 1 < len(layers):
            seq.append(nn.ReLU(inplace=True))
    return nn.Sequential(*seq)


class Encoder(nn.Module):
    def __init__(self, config, is_vae=False):
        super().__init__()
        self.is_vae = bool(is_vae)
        self.output_size = config['output_size']
        self.use_bias = config['use_bias']
        self.relu_slope = config['relu_slope']       # carried, never used: the stack is plain ReLU
        if not self.use_bias:
            raise NotImplementedError("the HIP encoder path expects use_bias=true (every reference config sets it)")

        pairs = zip(POINT_WIDTHS[:-1], POINT_WIDTHS[1:])
        self.conv = _interleave_relu([nn.Conv1d(cin, cout, 1, bias=True) for cin, cout in pairs], after_last=False)
        self.fc = _interleave_relu([nn.Linear(POOLED, POOLED)], after_last=True)
        for head in ("mu_layer", "std_layer"):       # std_layer exists (and is saved) even when is_vae is False
            setattr(self, head, nn.Linear(POOLED, self.output_size))

    def _params(self):
        """Flat parameter order of the C entry points: conv weights, conv biases, fc, mu[, std]."""
        convs = [m for m in self.conv if isinstance(m, nn.Conv1d)]
        tail = [self.fc[0], self.mu_layer] + ([self.std_layer] if self.is_vae else [])
        # a plain encoder never touches std_layer: its .grad stays None
        return [c.weight for c in convs] + [c.bias for c in convs] + [t for m in tail for t in (m.weight, m.bias)]

    def forward(self, x, eps=None):
        """x is channels-first (B, 3, N) — normally the transposed view FullModel makes of a (B, N, 3) batch, for
        which the ``contiguous()`` below is free.  ``eps`` injects the VAE draw (tests); otherwise device RNG."""
        pts = x.transpose(1, 2).contiguous()
        if not self.is_vae:
            return EncoderFunction.apply(pts, None, self.output_size, *self._params())
        if eps is None:
            eps = torch.randn((pts.size(0), self.output_size), dtype=torch.float32, device=pts.device)
        return EncoderFunction.apply(pts, eps.contiguous(), self.output_size, *self._params())
