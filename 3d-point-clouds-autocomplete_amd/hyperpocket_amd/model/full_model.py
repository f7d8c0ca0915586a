"""Drop-in for model/full_model.py:13-152 (FullModel + the HyperPocket / HyperRec / HyperCloud modes).

Same constructor config, ``forward(existing, missing, gt_shape, epoch, device, noise=None)``
signature, return convention (train: (reconstruction (B,3,N), exp(logvar), mu); eval:
reconstruction), mode-filtered ``parameters()``, ``get_noise_size()`` and the caller-visible side
effects of the reference (in-place transposes of ``existing``/``missing`` and the mutation of the
``gt_shape`` list) — so core/epoch_loops.py runs on it unmodified.

What changed underneath: the B-iteration Python loop over per-cloud TargetNetwork modules and CPU
point draws (model/full_model.py:70-74) is one batched launch sequence; every dense op is a HIP
kernel (ops.py).  Two keyword-only extras, ``points=`` and ``eps=``, let tests inject the random
draws the reference takes from its RNGs.
"""
from itertools import chain
from typing import Iterator

import os

import torch
import torch.nn as nn
from torch.nn import Parameter

from .encoder import Encoder
from .hyper_network import HyperNetwork
from .target_network import target_network_batched
from ..utils.points import generate_points, sample_points_device


class FullModel(nn.Module):

    @staticmethod
    def _complete_config(config):
        config['hyper_network']['target_network_layer_out_channels'] = config['target_network']['layer_out_channels']
        config['hyper_network']['target_network_use_bias'] = config['target_network']['use_bias']
        config['hyper_network']['input_size'] = config['random_encoder']['output_size'] + \
                                                config['real_encoder']['output_size']

        config['hyper_network']['target_network_freeze_layers_learning'] = config['target_network'][
            'freeze_layers_learning']

    def get_noise_size(self):
        return self.random_encoder_output_size

    def _resolve_mode(self, config):
        self.random_encoder_output_size = config['random_encoder']['output_size']
        if config['random_encoder']['output_size'] > 0 and config['real_encoder']['output_size'] > 0:
            self.mode = HyperPocket()
            self.random_encoder = Encoder(config['random_encoder'], is_vae=True)
            self.real_encoder = Encoder(config['real_encoder'], is_vae=False)
        elif config['random_encoder']['output_size'] > 0:
            self.mode = HyperCloud()
            self.random_encoder = Encoder(config['random_encoder'], is_vae=True)
        elif config['real_encoder']['output_size'] > 0:
            self.mode = HyperRec()
            self.real_encoder = Encoder(config['real_encoder'], is_vae=False)
        else:
            raise ValueError("at least one encoder should have non zero output")

    def __init__(self, config):
        super().__init__()
        self._complete_config(config)
        self._resolve_mode(config)

        self.hyper_network = HyperNetwork(config['hyper_network'])
        self.target_network_config = config['target_network']

        self.point_generator_config = {'target_network_input': config['target_network_input']}
        # 'device': one Philox launch per step (fast path); 'reference': the reference's per-cloud CPU
        # draws from the torch global generator, value for value (utils/points.py)
        self.point_sampler = 'device'
        self.concurrent_encoders = True   # HyperPocket training: run the two independent encoders on two streams
        # ... and their conv stacks as batched launches (one node for both encoders); HP_PAIRED_ENCODERS=0: two nodes
        self.paired_encoders = os.environ.get("HP_PAIRED_ENCODERS", "1") != "0"
        self._sampler_seed = None
        self._sampler_calls = 0

    def _draw_points(self, epoch, batch, n, device):
        if self.point_sampler == 'reference':
            pts = [generate_points(config=self.point_generator_config, epoch=epoch, size=(n, 3)) for _ in range(batch)]
            return torch.stack(pts).to(device)
        if self._sampler_seed is None:
            self._sampler_seed = torch.initial_seed()
        self._sampler_calls += 1
        return sample_points_device(self.point_generator_config, epoch, batch, n, device, self._sampler_seed,
                                    self._sampler_calls)

    def forward(self, existing, missing, gt_shape, epoch, device, noise=None, *, points=None, eps=None):

        if existing.size(-1) == 3:
            existing.transpose_(existing.dim() - 2, existing.dim() - 1)

        if noise is None and missing is not None and missing.size(-1) == 3:
            missing.transpose_(missing.dim() - 2, missing.dim() - 1)

        if gt_shape[-1] == 3:
            gt_shape[1], gt_shape[2] = gt_shape[2], gt_shape[1]

        latent, mu, logvar = self.mode.get_latent(self, existing, missing, noise, eps)
        self._last_latent = latent   # TrainEngine hooks its gradient: "hypernetwork backward has been enqueued"
        hook = self.__dict__.get("_pre_hypernet_hook")
        if hook is not None:
            hook()                   # TrainEngine: deferred all-reduce wait + Adam of the hypernetwork heads

        target_networks_weights = self.hyper_network(latent)
        batch, n_points = target_networks_weights.size(0), gt_shape[2]
        if points is None:
            points = self._draw_points(epoch, batch, n_points, latent.device)
        out = target_network_batched(self.target_network_config, target_networks_weights, points)   # (B, N, 3)

        # reconstruction shape [BATCH_SIZE, 3, N] — a transposed view of the (B, N, 3) kernel output, so the
        # caller's `reconstruction.permute(0, 2, 1)` (core/epoch_loops.py:26) is contiguous again for free
        reconstruction = out.permute(0, 2, 1)
        if self.training:
            return reconstruction, logvar, mu
        else:
            return reconstruction  # , latent, target_networks_weights

    def parameters(self, recurse: bool = True) -> Iterator[Parameter]:
        return self.mode.get_parameters(self)


class ModelMode(object):

    def get_latent(self, model: FullModel, existing, missing, noise=None, eps=None):
        raise NotImplementedError

    def get_parameters(self, model: FullModel) -> Iterator[Parameter]:
        raise NotImplementedError

    def has_generativity(self) -> bool:
        raise NotImplementedError


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


class HyperPocket(ModelMode):

    def get_latent(self, model: FullModel, existing, missing, noise=None, eps=None):
        if model.training:
            if (model.paired_encoders and missing.is_cuda and missing.shape == existing.shape
                    and model.random_encoder.output_size == model.real_encoder.output_size):
                # both conv stacks as batched launches, one autograd node (ops.EncoderPairFunction)
                from ..ops import EncoderPairFunction
                re, pe = model.random_encoder, model.real_encoder
                x0, x1 = missing.transpose(1, 2).contiguous(), existing.transpose(1, 2).contiguous()
                if eps is None:
                    eps = torch.randn((x0.size(0), re.output_size), dtype=torch.float32, device=x0.device)
                side = _side_stream(model, missing.device) if model.concurrent_encoders else None
                latent, mu, logvar = EncoderPairFunction.apply(x0, eps.contiguous(), x1, re.output_size, side,
                                                               model.__dict__.get("_after_encoder_tails"),
                                                               *re._params(), *pe._params())
                return latent, mu, logvar                    # latent = [codes | real_mu], written in place by the two encoders
            elif model.concurrent_encoders and missing.is_cuda:
                cur = torch.cuda.current_stream(missing.device)
                side = _side_stream(model, missing.device)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    codes, mu, logvar = model.random_encoder(missing, eps)
                real_mu = model.real_encoder(existing)
                cur.wait_stream(side)
                for t in (codes, mu, logvar):
                    t.record_stream(cur)
            else:
                codes, mu, logvar = model.random_encoder(missing, eps)
                real_mu = model.real_encoder(existing)
            latent = torch.cat([codes, real_mu], 1)
            return latent, mu, logvar
        else:
            if noise is None:
                _, random_mu, _ = model.random_encoder(missing, eps)
            else:
                random_mu = noise
            real_mu = model.real_encoder(existing)
            latent = torch.cat([random_mu, real_mu], 1)
            return latent, None, None

    def get_parameters(self, model: FullModel):
        return chain(model.random_encoder.parameters(),
                     model.real_encoder.parameters(),
                     model.hyper_network.parameters())

    def has_generativity(self) -> bool:
        return True


class HyperRec(ModelMode):

    def get_latent(self, model: FullModel, existing, missing, noise=None, eps=None):
        return model.real_encoder(existing), None, None

    def get_parameters(self, model: FullModel):
        return chain(model.real_encoder.parameters(), model.hyper_network.parameters())

    def has_generativity(self) -> bool:
        return False


class HyperCloud(ModelMode):

    def get_latent(self, model: FullModel, existing, missing, noise=None, eps=None):
        if model.training:
            return model.random_encoder(existing, eps)
        else:
            if noise is None:
                _, random_mu, _ = model.random_encoder(existing, eps)
            else:
                random_mu = noise
            return random_mu, None, None

    def get_parameters(self, model: FullModel):
        return chain(model.random_encoder.parameters(), model.hyper_network.parameters())

    def has_generativity(self) -> bool:
        return False
