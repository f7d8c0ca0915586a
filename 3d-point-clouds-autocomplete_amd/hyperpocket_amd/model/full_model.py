"""FullModel: encoders -> latent -> hypernetwork -> batched target network, as one launch sequence.

The interface is the reference's (model/full_model.py:13-152), because core/epoch_loops.py must run on it
unmodified: ``FullModel(config)``, ``forward(existing, missing, gt_shape, epoch, device, noise=None)`` returning
``(reconstruction (B,3,N), exp(logvar), mu)`` in training and ``reconstruction`` in eval; ``parameters()`` limited
to the sub-networks the active mode trains; ``get_noise_size()``; ``mode.has_generativity()``; submodule names
``random_encoder`` / ``real_encoder`` / ``hyper_network`` (state_dict keys); the keys the constructor adds to
``config['hyper_network']``; and the caller-visible side effects — ``existing`` / ``missing`` transposed in place
and the ``gt_shape`` list permuted.

Underneath nothing is shared with it: the three operating modes are rows of a table rather than classes, and
the reference's loop over B per-cloud TargetNetwork modules with CPU point draws (:70-74) is one batched kernel
sequence (ops.py).  Keyword-only ``points=`` / ``eps=`` let tests inject the draws the reference takes from its
RNGs.
"""
import os
from dataclasses import dataclass
from typing import Optional

import torch
from torch import nn

from .encoder import Encoder
from .hyper_network import HyperNetwork
from .target_network import target_network_batched
from ..utils.points import generate_points, sample_points_device


@dataclass(frozen=True)
class Mode:
    """One operating mode.  ``vae_input`` names the forward argument the VAE ("random") encoder reads, or None when
    the mode has no VAE encoder; ``conditioned`` says whether the plain ("real") encoder of ``existing`` contributes
    the second half of the latent; ``kld`` whether the training loss carries the KL term."""
    name: str
    vae_input: Optional[str]
    conditioned: bool
    kld: bool

    def has_generativity(self) -> bool:          # the name core/epoch_loops.py:28 asks for
        return self.kld

    @property
    def trained(self):
        """Sub-networks whose parameters the optimiser sees, in flat-buffer order."""
        enc = (("random_encoder",) if self.vae_input else ()) + (("real_encoder",) if self.conditioned else ())
        return enc + ("hyper_network",)


#            (random size > 0, real size > 0) -> mode
MODES = {(True, True): Mode("HyperPocket", "missing", True, True),     # complete `existing` with a sampled part
         (True, False): Mode("HyperCloud", "existing", False, False),  # VAE over whole clouds, loss without KLD
         (False, True): Mode("HyperRec", None, True, False)}           # deterministic reconstruction


def _side_stream(model, device):
    """One extra HIP stream per model: the two encoders of a HyperPocket step are independent, so the VAE encoder runs
    there while the plain one runs on the caller's stream (their many small head kernels then hide under the other's
    wide GEMMs, forward and — autograd replays each node on its forward stream — backward)."""
    streams = model.__dict__.setdefault("_side_streams", {})
    key = (device.type, device.index)
    if key not in streams:
        # high priority: ROCclr keeps a separate pool of hardware queues per priority, so this stream can never be
        # folded onto the hardware queue of the caller's (normal-priority) stream.  It is at GPU_MAX_HW_QUEUES=4 (the
        # default) once a process group's streams exist: normal-pool streams then share queues, the two encoders
        # serialise and a step loses 0.45 ms (measured; tools/micro/stream_queues.py shows the mapping).
        streams[key] = torch.cuda.Stream(device=device, priority=-1)
    return streams[key]


def _channels_first_(t):
    """In-place (..., N, 3) -> (..., 3, N) view swap, as the reference does to its inputs; the caller sees it."""
    if t is not None and t.size(-1) == 3:
        t.transpose_(-2, -1)


class FullModel(nn.Module):
    def __init__(self, config):
        super().__init__()
        sizes = {k: config[k]['output_size'] for k in ("random_encoder", "real_encoder")}
        tn = config['target_network']
        # the hypernetwork's config is completed in place; callers (and saved configs) see these keys afterwards
        config['hyper_network'].update(
            target_network_layer_out_channels=tn['layer_out_channels'],
            target_network_use_bias=tn['use_bias'],
            input_size=sizes["random_encoder"] + sizes["real_encoder"],
            target_network_freeze_layers_learning=tn['freeze_layers_learning'])

        self.mode = MODES.get((sizes["random_encoder"] > 0, sizes["real_encoder"] > 0))
        if self.mode is None:
            raise ValueError("at least one encoder should have non zero output")
        self.random_encoder_output_size = sizes["random_encoder"]
        for name in self.mode.trained[:-1]:       # VAE encoder first: registration (= checkpoint and RNG) order
            setattr(self, name, Encoder(config[name], is_vae=(name == "random_encoder")))
        self.hyper_network = HyperNetwork(config['hyper_network'])
        self.target_network_config = tn
        self.point_generator_config = {'target_network_input': config['target_network_input']}

        # 'device': one Philox launch per step (fast path); 'reference': the reference's per-cloud CPU
        # draws from the torch global generator, value for value (utils/points.py)
        self.point_sampler = 'device'
        self.concurrent_encoders = True   # HyperPocket training: run the two independent encoders on two streams
        # ... and their conv stacks as batched launches (one node for both encoders); HP_PAIRED_ENCODERS=0: two nodes
        self.paired_encoders = os.environ.get("HP_PAIRED_ENCODERS", "1") != "0"
        self._sampler_seed = None
        self._sampler_calls = 0

    # ---- what the caller's loop asks of the model ----------------------------------------------------------------
    def get_noise_size(self):
        return self.random_encoder_output_size

    def parameters(self, recurse: bool = True):
        for name in self.mode.trained:
            yield from getattr(self, name).parameters()

    # ---- latent ----------------------------------------------------------------------------------------------------
    def _pair_latent(self, existing, missing, eps):
        """HyperPocket training, equal-sized halves: both conv stacks as batched launches under one autograd node;
        the two encoders write [z | real_mu] straight into the latent."""
        from ..ops import EncoderPairFunction
        vae, plain = self.random_encoder, self.real_encoder
        x_vae, x_plain = missing.transpose(1, 2).contiguous(), existing.transpose(1, 2).contiguous()
        if eps is None:
            eps = torch.randn((x_vae.size(0), vae.output_size), dtype=torch.float32, device=x_vae.device)
        side = _side_stream(self, missing.device) if self.concurrent_encoders else None
        return EncoderPairFunction.apply(x_vae, eps.contiguous(), x_plain, vae.output_size, side,
                                         self.__dict__.get("_after_encoder_tails"),
                                         *vae._params(), *plain._params())

    def _two_stream_latent(self, existing, missing, eps):
        """HyperPocket training, general shapes: VAE encoder on the side stream, plain encoder on the caller's."""
        cur = torch.cuda.current_stream(missing.device)
        side = _side_stream(self, missing.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            z, mu, var = self.random_encoder(missing, eps)
        cond = self.real_encoder(existing)
        cur.wait_stream(side)
        for t in (z, mu, var):
            t.record_stream(cur)
        return torch.cat([z, cond], 1), mu, var

    def _latent(self, existing, missing, noise, eps):
        """-> (latent, mu, exp(logvar)); the last two are None outside VAE training."""
        mode, parts, mu, var = self.mode, [], None, None
        if mode.vae_input:
            source = missing if mode.vae_input == "missing" else existing
            if self.training:
                if mode.conditioned and source.is_cuda:
                    same = (source.shape == existing.shape
                            and self.random_encoder.output_size == self.real_encoder.output_size)
                    if self.paired_encoders and same:
                        return self._pair_latent(existing, source, eps)
                    if self.concurrent_encoders:
                        return self._two_stream_latent(existing, source, eps)
                z, mu, var = self.random_encoder(source, eps)
                parts.append(z)
            elif noise is not None:
                parts.append(noise)                                # experiments: a fixed latent half, encoder skipped
            else:
                parts.append(self.random_encoder(source, eps)[1])  # eval: the posterior mean
        if mode.conditioned:
            parts.append(self.real_encoder(existing))
        return (parts[0] if len(parts) == 1 else torch.cat(parts, 1)), mu, var

    # ---- decoder input ---------------------------------------------------------------------------------------------
    def _draw_points(self, epoch, batch, n, device):
        if self.point_sampler == 'reference':
            draws = [generate_points(config=self.point_generator_config, epoch=epoch, size=(n, 3))
                     for _ in range(batch)]
            return torch.stack(draws).to(device)
        if self._sampler_seed is None:
            self._sampler_seed = torch.initial_seed()
        self._sampler_calls += 1
        return sample_points_device(self.point_generator_config, epoch, batch, n, device, self._sampler_seed,
                                    self._sampler_calls)

    def forward(self, existing, missing, gt_shape, epoch, device, noise=None, *, points=None, eps=None):
        _channels_first_(existing)
        if noise is None:
            _channels_first_(missing)
        if gt_shape[-1] == 3:                     # the caller's list ends up (B, 3, N) too
            gt_shape[1:3] = gt_shape[2], gt_shape[1]

        latent, mu, var = self._latent(existing, missing, noise, eps)
        self._last_latent = latent   # TrainEngine hooks its gradient: "hypernetwork backward has been enqueued"
        hook = self.__dict__.get("_pre_hypernet_hook")
        if hook is not None:
            hook()                   # TrainEngine: deferred all-reduce wait + Adam of the hypernetwork heads

        theta = self.hyper_network(latent)        # (B, T): one target network per cloud
        if points is None:
            points = self._draw_points(epoch, theta.size(0), gt_shape[2], latent.device)
        cloud = target_network_batched(self.target_network_config, theta, points)      # (B, N, 3)
        # handed out as the (B, 3, N) view: the caller's `.permute(0, 2, 1)` (core/epoch_loops.py:26) then lands on
        # contiguous memory again, no copy either way
        reconstruction = cloud.permute(0, 2, 1)
        return (reconstruction, var, mu) if self.training else reconstruction
