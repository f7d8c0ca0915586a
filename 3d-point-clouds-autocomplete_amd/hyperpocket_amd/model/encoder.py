"""Drop-in for model/encoder.py:5-53 — same constructor, parameter names/shapes (state_dict
compatible: conv.{0,2,4,6,8}, fc.0, mu_layer, std_layer) and return convention; the compute is
hp_encoder_forward / hp_encoder_backward (fp32 MFMA GEMM stack + fused max-pool bookkeeping)."""
import torch
import torch.nn as nn

from ..ops import EncoderFunction


class Encoder(nn.Module):
    def __init__(self, config, is_vae=False):
        super().__init__()

        self.output_size = config['output_size']
        self.use_bias = config['use_bias']
        self.relu_slope = config['relu_slope']   # read but unused, as in the reference
        self.is_vae = is_vae
        if not self.use_bias:
            raise NotImplementedError("the HIP encoder path expects use_bias=true (every reference config sets it)")

        # Parameter containers only (same registration order and default-init RNG consumption as the
        # reference, so a seeded construction yields identical weights).  No BatchNorm.
        self.conv = nn.Sequential(
            nn.Conv1d(in_channels=3, out_channels=64, kernel_size=1, bias=self.use_bias),
            nn.ReLU(inplace=True),

            nn.Conv1d(in_channels=64, out_channels=128, kernel_size=1, bias=self.use_bias),
            nn.ReLU(inplace=True),

            nn.Conv1d(in_channels=128, out_channels=256, kernel_size=1, bias=self.use_bias),
            nn.ReLU(inplace=True),

            nn.Conv1d(in_channels=256, out_channels=512, kernel_size=1, bias=self.use_bias),
            nn.ReLU(inplace=True),

            nn.Conv1d(in_channels=512, out_channels=512, kernel_size=1, bias=self.use_bias),
        )

        self.fc = nn.Sequential(
            nn.Linear(512, 512, bias=True),
            nn.ReLU(inplace=True)
        )

        self.mu_layer = nn.Linear(512, self.output_size, bias=True)
        self.std_layer = nn.Linear(512, self.output_size, bias=True)

    def _params(self):
        convs = [self.conv[i] for i in (0, 2, 4, 6, 8)]
        ps = [c.weight for c in convs] + [c.bias for c in convs] + \
             [self.fc[0].weight, self.fc[0].bias, self.mu_layer.weight, self.mu_layer.bias]
        if self.is_vae:   # std_layer is never touched by a non-VAE encoder: its .grad stays None
            ps += [self.std_layer.weight, self.std_layer.bias]
        return ps

    def forward(self, x, eps=None):
        """x: (B, 3, N) as in the reference (typically the transposed *view* FullModel makes of a
        (B, N, 3) batch — then no copy happens here), or (B, N, 3) when x.size(-1) == 3 is unambiguous
        is NOT assumed: the reference contract is channels-first."""
        pts = x.transpose(1, 2).contiguous()          # (B, N, 3) rows = points; free for a transposed view
        if self.is_vae:
            if eps is None:
                # model/encoder.py:40 eps = torch.randn_like(std) — device RNG, plumbing
                eps = torch.randn((pts.size(0), self.output_size), dtype=torch.float32, device=pts.device)
            return EncoderFunction.apply(pts, eps.contiguous(), self.output_size, *self._params())
        return EncoderFunction.apply(pts, None, self.output_size, *self._params())
