"""autograd bridges between PyTorch tensors and the model entry points of the C ABI.

Each Function's forward/backward is ONE call into libhyperpocket_hip.so (which issues the whole
launch sequence on the current stream); PyTorch only owns the memory and the autograd wiring.
"""
import contextlib
import ctypes
from ctypes import c_int, c_long, c_void_p

import torch
from torch.autograd import Function

from ._lib import HipExtensionError, call, check_input, current_stream, load_library, ptr

HP_MAX_HEADS = 8


class _EncoderPtrs(ctypes.Structure):  # HpEncoderWeights / HpEncoderGrads (csrc/hp_model.h)
    _fields_ = [("conv_w", c_void_p * 5), ("conv_b", c_void_p * 5), ("fc_w", c_void_p), ("fc_b", c_void_p),
                ("mu_w", c_void_p), ("mu_b", c_void_p), ("std_w", c_void_p), ("std_b", c_void_p)]


class _EncoderIO(ctypes.Structure):  # HpEncoderIO (one encoder's buffers for hp_encoder_forward_pair)
    _fields_ = [("x", c_void_p), ("w", ctypes.POINTER(_EncoderPtrs)), ("eps", c_void_p), ("argidx", c_void_p),
                ("g", c_void_p), ("f", c_void_p), ("mu", c_void_p), ("lv", c_void_p), ("z", c_void_p), ("explv", c_void_p),
                ("ws", c_void_p), ("is_vae", c_int), ("out_ld", c_int)]


class _EncoderBwdIO(ctypes.Structure):  # HpEncoderBwdIO (one encoder's buffers for hp_encoder_backward_pair)
    _fields_ = [("x", c_void_p), ("w", ctypes.POINTER(_EncoderPtrs)), ("eps", c_void_p), ("argidx", c_void_p),
                ("g", c_void_p), ("f", c_void_p), ("lv", c_void_p), ("grad_out", c_void_p), ("grad_mu", c_void_p),
                ("grad_explv", c_void_p), ("gr", ctypes.POINTER(_EncoderPtrs)), ("ws", c_void_p), ("fwd_ws", c_void_p),
                ("is_vae", c_int), ("grad_out_ld", c_int)]


class _HyperWeights(ctypes.Structure):  # HpHyperWeights
    _fields_ = [("trunk_w", c_void_p * 5), ("trunk_b", c_void_p * 5), ("n_heads", c_int),
                ("head_out", c_int * HP_MAX_HEADS), ("head_w", c_void_p * HP_MAX_HEADS),
                ("head_b", c_void_p * HP_MAX_HEADS)]


class _HyperGrads(ctypes.Structure):  # HpHyperGrads
    _fields_ = [("trunk_w", c_void_p * 5), ("trunk_b", c_void_p * 5), ("head_w", c_void_p * HP_MAX_HEADS),
                ("head_b", c_void_p * HP_MAX_HEADS)]


def _dp(t):
    return None if t is None else t.data_ptr()


def _long_fn(name, *args):
    fn = getattr(load_library(), name)
    fn.restype = c_long
    return fn(*args)


# ---------------------------------------------------------------------------------------------
# Arithmetic: by default the wide GEMM-shaped kernels form each fp32 product on the f16 / bf16 matrix pipe from
# exact pieces of the fp32 operands (fp32 accumulation; see bench.py config.arithmetic).  Every one of them has an
# IEEE-fp32 form (v_mfma_f32_* / VALU fma) behind a process-wide switch of the library.
# ---------------------------------------------------------------------------------------------
_PIECE_SWITCHES = ("hp_conv_split_set",                   # encoder conv stack: two f16 pieces per operand
                   "hp_conv_presplit_set",                # ... hidden activations stored as piece pairs
                   "hp_encoder_backward_set_chain_f16",   # encoder backward: delta chain + dW launch
                   "hp_hypernet_set_heads_stream",        # hypernetwork heads' forward: three bf16 pieces
                   "hp_target_fused_set_f16")             # fused decoder forward


@contextlib.contextmanager
def strict_fp32():
    """Inside the block every kernel computes its products in fp32 (no f16 / bf16 pieces): the arithmetic the
    reference states.  The switches are process-wide (not per stream or thread) and restored on exit."""
    lib = load_library()
    was = [getattr(lib, name)(0) for name in _PIECE_SWITCHES]
    try:
        yield
    finally:
        for name, prev in zip(_PIECE_SWITCHES, was):
            getattr(lib, name)(prev)


# ---------------------------------------------------------------------------------------------
# Gradient placement: a FlatParameters owner (parallel.py) may register, per parameter, a view of
# its flat gradient buffer; backward then writes gradients straight into it (no copy before the
# RCCL all-reduce / fused Adam).  Default: fresh tensors.
# ---------------------------------------------------------------------------------------------
_GRAD_VIEWS = {}


def register_grad_view(param, view):
    """Returns the registry key; the owner drops its keys when it dies (parallel.FlatParameters does, through a
    weakref finalizer), so a dead engine's flat gradient buffer is not pinned by this table."""
    _GRAD_VIEWS[param.data_ptr()] = view
    return param.data_ptr()


def drop_grad_views(keys):
    for k in keys:
        _GRAD_VIEWS.pop(k, None)


def clear_grad_views():
    _GRAD_VIEWS.clear()


def _grad_buffer(param):
    v = _GRAD_VIEWS.get(param.data_ptr())
    if v is not None and v.shape == param.shape:
        # a FRESH view object: autograd's AccumulateGrad keeps (instead of cloning) a gradient nobody else references,
        # so param.grad ends up aliasing the flat buffer with no copy
        return v.view(v.shape)
    return torch.empty_like(param, memory_format=torch.contiguous_format)


# The encoder backward copies the critical rows' activations out of the forward's workspace (15 KB per point, kept
# alive by the autograd node).  False: drop the workspace after the forward and recompute those rows instead.
KEEP_ENCODER_ACTIVATIONS = True
# Channels whose max-pool peaks at the same point share every activation below it: the encoder backward runs its
# layers 4..1 on the distinct critical points (~170 of 512 per cloud).  False: one row per (cloud, channel).
DEDUP_CRITICAL_ROWS = True
# The two encoders of a HyperPocket step share the launches of their backward (hp_encoder_backward_pair, one stream).
# False: two hp_encoder_backward_ld calls on two streams (round 2's form).
PAIRED_ENCODER_BACKWARD = True


def _encoder_struct(params, cls=_EncoderPtrs):
    # params: conv_w x5, conv_b x5, fc_w, fc_b, mu_w, mu_b[, std_w, std_b]
    s = cls()
    for i in range(5):
        s.conv_w[i] = _dp(params[i])
        s.conv_b[i] = _dp(params[5 + i])
    s.fc_w, s.fc_b, s.mu_w, s.mu_b = (_dp(p) for p in params[10:14])
    if len(params) > 14:
        s.std_w, s.std_b = _dp(params[14]), _dp(params[15])
    return s


class EncoderFunction(Function):
    """model/encoder.py:43-53.  x: (B, Np, 3) contiguous.  params as listed in _encoder_struct.
    Returns mu (plain) or (z, mu, exp(logvar)) (VAE)."""

    @staticmethod
    def forward(ctx, x, eps, out_size, *params):
        check_input(x, "x")
        is_vae = len(params) == 16
        for i, p in enumerate(params):
            check_input(p, f"encoder param {i}")
        B, Np = x.size(0), x.size(1)
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        argidx = torch.empty((B, 512), dtype=torch.int32, device=dev)
        g = torch.empty((B, 512), **f32)
        f = torch.empty((B, 512), **f32)
        mu = torch.empty((B, out_size), **f32)
        lv = z = explv = None
        if is_vae:
            check_input(eps, "eps")
            lv, z, explv = (torch.empty((B, out_size), **f32) for _ in range(3))
        ws = torch.empty((_long_fn("hp_encoder_forward_workspace_floats", B, Np),), **f32)
        w = _encoder_struct(params)
        call("hp_encoder_forward", B, Np, x, ctypes.byref(w), out_size, int(is_vae), eps, argidx, g, f, mu, lv, z, explv,
             ws, current_stream(dev))
        ctx.is_vae, ctx.out_size = is_vae, out_size
        ctx.fwd_ws = ws if KEEP_ENCODER_ACTIVATIONS else None
        ctx.save_for_backward(x, eps, argidx, g, f, lv, *params)
        if is_vae:
            return z, mu, explv
        return mu

    @staticmethod
    def backward(ctx, *grads):
        x, eps, argidx, g, f, lv, *params = ctx.saved_tensors
        B, Np = x.size(0), x.size(1)
        dev = x.device
        if ctx.is_vae:
            gz, gmu, gexplv = (None if t is None else t.contiguous() for t in grads)
            gout = gz
        else:
            gout, gmu, gexplv = grads[0].contiguous(), None, None
        out = [_grad_buffer(p) for p in params]
        ws = torch.empty((_long_fn("hp_encoder_backward_workspace_floats", B, ctx.out_size),), dtype=torch.float32,
                         device=dev)
        w, gr = _encoder_struct(params), _encoder_struct(out)
        call("hp_encoder_backward", B, Np, x, ctypes.byref(w), ctx.out_size, int(ctx.is_vae), eps, argidx, g, f, lv,
             gout, gmu, gexplv, ctypes.byref(gr), ws, ctx.fwd_ws, int(DEDUP_CRITICAL_ROWS), current_stream(dev))
        ctx.fwd_ws = None
        return (None, None, None, *out)


class EncoderPairFunction(Function):
    """The two encoders of a HyperPocket training step (model/full_model.py:106-112) as ONE node: the conv stacks of both run
    as batched launches (hp_encoder_forward_pair), and so do their backward's (hp_encoder_backward_pair).  Arguments: x_vae (missing), eps, x_plain (existing), out_size, side stream, then the VAE encoder's 16
    parameters and the plain encoder's 14.  Returns (latent, mu, exp(logvar)) with latent = [z | real_mu] (B, 2*out): the two
    encoders write its halves directly (no torch.cat) and the backward reads the halves of d latent in place."""

    N_VAE = 16

    @staticmethod
    def forward(ctx, x0, eps, x1, out_size, side, after_tails, *params):
        p0, p1 = params[:EncoderPairFunction.N_VAE], params[EncoderPairFunction.N_VAE:]
        check_input(x0, "x (VAE encoder)")
        check_input(x1, "x (plain encoder)")
        check_input(eps, "eps")
        for i, p in enumerate(params):
            check_input(p, f"encoder param {i}")
        if x0.shape != x1.shape or len(p1) != 14:
            raise HipExtensionError("EncoderPairFunction: both encoders must see the same (B, N, 3) shape")
        B, Np = x0.size(0), x0.size(1)
        dev = x0.device
        f32 = dict(dtype=torch.float32, device=dev)
        nws = _long_fn("hp_encoder_forward_workspace_floats", B, Np)
        io = (_EncoderIO * 2)()
        keep, structs = [], []
        latent = torch.empty((B, 2 * out_size), **f32)
        for e, (x, ps, vae) in enumerate(((x0, p0, True), (x1, p1, False))):
            argidx = torch.empty((B, 512), dtype=torch.int32, device=dev)
            g, f = torch.empty((B, 512), **f32), torch.empty((B, 512), **f32)
            lv = z = explv = None
            if vae:
                mu = torch.empty((B, out_size), **f32)
                lv, explv = (torch.empty((B, out_size), **f32) for _ in range(2))
                z = latent                                   # columns [0, out)
            else:
                mu = latent[:, out_size:]                    # columns [out, 2*out): data_ptr() is the block's first element
            ws = torch.empty((nws,), **f32)
            w = _encoder_struct(ps)
            structs.append(w)
            io[e].x, io[e].w, io[e].eps, io[e].argidx = x.data_ptr(), ctypes.pointer(w), _dp(eps if vae else None), argidx.data_ptr()
            io[e].g, io[e].f, io[e].mu, io[e].lv, io[e].z, io[e].explv = (_dp(t) for t in (g, f, mu, lv, z, explv))
            io[e].ws, io[e].is_vae, io[e].out_ld = ws.data_ptr(), int(vae), 2 * out_size
            keep.append((argidx, g, f, mu, lv, z, explv, ws))
        call("hp_encoder_forward_pair", B, Np, out_size, io, current_stream(dev))
        ctx.out_size, ctx.side, ctx.after_tails = out_size, side, after_tails
        ctx.fwd_ws = [k[7] if KEEP_ENCODER_ACTIVATIONS else None for k in keep]
        ctx.save_for_backward(x0, eps, x1, keep[0][0], keep[0][1], keep[0][2], keep[0][4], keep[1][0], keep[1][1], keep[1][2],
                              *params)
        return latent, keep[0][3], keep[0][6]

    @staticmethod
    def backward(ctx, glat, gmu, gexplv):
        x0, eps, x1, arg0, g0, f0, lv0, arg1, g1, f1, *params = ctx.saved_tensors
        p0, p1 = params[:EncoderPairFunction.N_VAE], params[EncoderPairFunction.N_VAE:]
        B, Np = x0.size(0), x0.size(1)
        dev = x0.device
        cur = torch.cuda.current_stream(dev)
        side = ctx.side if ctx.side is not None else cur
        nws = _long_fn("hp_encoder_backward_workspace_floats", B, ctx.out_size)
        glat, gmu, gexplv = (None if t is None else t.contiguous() for t in (glat, gmu, gexplv))
        if glat is None:
            glat = torch.zeros((B, 2 * ctx.out_size), dtype=torch.float32, device=dev)
        out0, out1 = [_grad_buffer(p) for p in p0], [_grad_buffer(p) for p in p1]
        gz, greal = glat, glat[:, ctx.out_size:]             # the halves of d latent, read in place (row stride 2*out)

        def run(x, ps, outs, vae, argidx, g, f, lv, gout, gm, ge, fwd_ws):
            ws = torch.empty((nws,), dtype=torch.float32, device=dev)
            w, gr = _encoder_struct(ps), _encoder_struct(outs)
            call("hp_encoder_backward_ld", B, Np, x, ctypes.byref(w), ctx.out_size, int(vae), eps if vae else None, argidx, g,
                 f, lv, ctypes.c_void_p(gout.data_ptr()), 2 * ctx.out_size, gm, ge, ctypes.byref(gr), ws, fwd_ws,
                 int(DEDUP_CRITICAL_ROWS), current_stream(dev))

        if PAIRED_ENCODER_BACKWARD:
            # one call, one stream: the conv stacks of both encoders in four shared launches, the tails in three
            io = (_EncoderBwdIO * 2)()
            keep = []
            for e, (x, ps, outs, vae, argidx, g, f, lv, gout, gm, ge, fwd_ws) in enumerate((
                    (x0, p0, out0, True, arg0, g0, f0, lv0, gz, gmu, gexplv, ctx.fwd_ws[0]),
                    (x1, p1, out1, False, arg1, g1, f1, None, greal, None, None, ctx.fwd_ws[1]))):
                ws = torch.empty((nws,), dtype=torch.float32, device=dev)
                w, gr = _encoder_struct(ps), _encoder_struct(outs)
                keep.append((ws, w, gr))
                io[e].x, io[e].w, io[e].eps, io[e].argidx = x.data_ptr(), ctypes.pointer(w), _dp(eps if vae else None), argidx.data_ptr()
                io[e].g, io[e].f, io[e].lv = g.data_ptr(), f.data_ptr(), _dp(lv)
                io[e].grad_out, io[e].grad_mu, io[e].grad_explv = gout.data_ptr(), _dp(gm), _dp(ge)
                io[e].gr, io[e].ws, io[e].fwd_ws = ctypes.pointer(gr), ws.data_ptr(), _dp(fwd_ws)
                io[e].is_vae, io[e].grad_out_ld = int(vae), 2 * ctx.out_size
            # `after_tails` (core/engine.py: the heads' fused dW + Adam pass): work of ANOTHER stream that should start
            # behind the two tails' three launches — an HBM-saturating pass that occupies every CU would otherwise hold
            # the tails' first launch back for its whole duration, while beside the matrix-bound conv-stack launches
            # it costs little.  The library orders that stream behind the tails (event record + stream wait).
            job, ctx.after_tails = ctx.after_tails, None
            if job is not None and not job.pending():
                job = None
            call("hp_encoder_backward_pair_ordered", B, Np, ctx.out_size, io, int(DEDUP_CRITICAL_ROWS), current_stream(dev),
                 ctypes.c_void_p(job.stream.cuda_stream) if job is not None else None)
            if job is not None:
                job.launch_ordered()
            ctx.fwd_ws = None
            return (None, None, None, None, None, None, *out0, *out1)
        # the two chains are independent: the VAE encoder's goes to the side stream
        if side is not cur:
            side.wait_stream(cur)
        with torch.cuda.stream(side):
            run(x0, p0, out0, True, arg0, g0, f0, lv0, gz, gmu, gexplv, ctx.fwd_ws[0])
        run(x1, p1, out1, False, arg1, g1, f1, None, greal, None, None, ctx.fwd_ws[1])
        if side is not cur:
            cur.wait_stream(side)
            for t in (glat, gmu, gexplv):
                if t is not None:
                    t.record_stream(side)
        ctx.fwd_ws = None
        return (None, None, None, None, None, None, *out0, *out1)


# The heads' weight gradient may be left to an exchange object with `accepts(head_weights) -> bool`, `begin(grad_theta, t5)`
# and optionally `finish(grad_theta, t5)` (core/engine.py: HeadsShard under data parallelism, FusedHeadsAdam on one GPU): the
# ranks then exchange the gradient's two factors (d theta, t5) instead of the 156 MB matrix / one kernel forms the gradient
# and applies Adam without storing it.  The object travels with the autograd node: HyperNetwork.forward reads it from its
# module (`hyper_network._heads_exchange`, set by the engine for the forward of the step it drives) and hands it to
# HyperNetFunction.apply — no process-global state, two engines in one process do not see each other's.


class HyperNetFunction(Function):
    """model/hyper_network.py:41-43.  params: trunk_w x5, trunk_b x5, head_w x H, head_b x H."""

    @staticmethod
    def forward(ctx, latent, n_heads, exchange, *params):
        latent = latent.contiguous()
        check_input(latent, "latent")
        # raw pointers go to the kernels: a head left on the CPU (`freeze_layers_learning` keeps `output` a plain list,
        # which .cuda() does not move — model/hyper_network.py:38-39), a .half() / .double() model or a foreign device must
        # fail here, not as a GPU memory fault
        for i, p in enumerate(params):
            check_input(p, f"hypernetwork param {i}")
            if p.device != latent.device:
                raise HipExtensionError(f"hypernetwork param {i} is on {p.device}, the latent on {latent.device}")
        B, in_size = latent.shape
        dev = latent.device
        w = _HyperWeights()
        for i in range(5):
            w.trunk_w[i], w.trunk_b[i] = _dp(params[i]), _dp(params[5 + i])
        w.n_heads = n_heads
        total = 0
        for h in range(n_heads):
            hw, hb = params[10 + h], params[10 + n_heads + h]
            w.head_w[h], w.head_b[h], w.head_out[h] = _dp(hw), _dp(hb), hw.size(0)
            total += hw.size(0)
        t = torch.empty((_long_fn("hp_hypernet_saved_floats", B),), dtype=torch.float32, device=dev)
        theta = torch.empty((B, total), dtype=torch.float32, device=dev)
        call("hp_hypernet_forward", B, in_size, latent, ctypes.byref(w), t, theta, total, current_stream(dev))
        ctx.n_heads, ctx.exchange = n_heads, exchange
        ctx.save_for_backward(latent, t, *params)
        return theta

    @staticmethod
    def backward(ctx, grad_theta):
        latent, t, *params = ctx.saved_tensors
        n_heads = ctx.n_heads
        grad_theta = grad_theta.contiguous()
        B, in_size = latent.shape
        dev = latent.device
        w, gr = _HyperWeights(), _HyperGrads()
        exch, ctx.exchange = ctx.exchange, None
        external_dw = exch is not None and exch.accepts(params[10:10 + n_heads])
        out = [None if external_dw and 10 <= i < 10 + n_heads else _grad_buffer(p) for i, p in enumerate(params)]
        for i in range(5):
            w.trunk_w[i], w.trunk_b[i] = _dp(params[i]), _dp(params[5 + i])
            gr.trunk_w[i], gr.trunk_b[i] = _dp(out[i]), _dp(out[5 + i])
        w.n_heads = n_heads
        for h in range(n_heads):
            w.head_w[h], w.head_b[h], w.head_out[h] = _dp(params[10 + h]), _dp(params[10 + n_heads + h]), params[10 + h].size(0)
            gr.head_w[h], gr.head_b[h] = _dp(out[10 + h]), _dp(out[10 + n_heads + h])
        if external_dw:
            # issued before this rank's backward launches, so the gathers travel under them
            o5 = _long_fn("hp_hypernet_t5_offset", B)
            exch.begin(grad_theta, t[o5:o5 + B * 2048].view(B, 2048))
        grad_latent = torch.empty_like(latent) if ctx.needs_input_grad[0] else None
        ws = torch.empty((_long_fn("hp_hypernet_backward_workspace_floats", B),), dtype=torch.float32, device=dev)
        # An exchange object with a stream of its own and `early` set (core/engine.py FusedHeadsAdam) gets that stream ordered
        # behind the LAST READER of the heads' weights inside the call (d t5 = d theta . W): its in-place dW + Adam pass then
        # starts beside the trunk's backward launches, on the CUs its persistent grid occupies.
        early = external_dw and getattr(exch, "early", False) and getattr(exch, "stream", None) is not None
        if early:
            call("hp_hypernet_backward_ordered", B, in_size, latent, ctypes.byref(w), t, grad_theta, grad_theta.size(1),
                 ctypes.byref(gr), grad_latent, ws, current_stream(dev), ctypes.c_void_p(exch.stream.cuda_stream))
        else:
            call("hp_hypernet_backward", B, in_size, latent, ctypes.byref(w), t, grad_theta, grad_theta.size(1), ctypes.byref(gr),
                 grad_latent, ws, current_stream(dev))
        if external_dw and hasattr(exch, "finish"):
            # in-place consumers of the heads' weights (the fused dW + Adam pass) go behind the backward that reads them
            o5 = _long_fn("hp_hypernet_t5_offset", B)
            exch.finish(grad_theta, t[o5:o5 + B * 2048].view(B, 2048))
        return (grad_latent, None, None, *out)


# The published decoder (3-32-64-128-64-3) runs as one fused kernel per direction (csrc/target_fused.hip: weights in
# LDS, activations in registers, nothing saved for the backward).  False: the layered batched-GEMM path, which serves
# every other architecture anyway.
FUSED_TARGET_NETWORK = True


class TargetNetworkFunction(Function):
    """All B per-cloud target networks at once (model/full_model.py:70-74 + model/target_network.py).
    theta (B, T), points (B, N, 3) -> y (B, N, 3)."""

    @staticmethod
    def forward(ctx, theta, points, channels):
        theta = theta.contiguous()
        points = points.contiguous()
        check_input(theta, "theta")
        check_input(points, "points")
        B, N = points.size(0), points.size(1)
        dev = theta.device
        ch = (c_int * len(channels))(*channels)
        need = _long_fn("hp_target_theta_size", len(channels), ch)
        if need != theta.size(1) or theta.size(0) != B:
            # model/target_network.py:29 `assert split_index == len(weights)`
            raise HipExtensionError(f"target network expects {need} weights per cloud, got {tuple(theta.shape)}")
        y = torch.empty((B, N, 3), dtype=torch.float32, device=dev)
        ctx.channels = tuple(channels)
        ctx.fused = bool(FUSED_TARGET_NETWORK and load_library().hp_target_fused_supported(len(channels), ch))
        if ctx.fused:
            call("hp_target_fused_forward", B, N, theta, theta.size(1), points, y, current_stream(dev))
            ctx.save_for_backward(theta, points)
            return y
        acts = torch.empty((_long_fn("hp_target_saved_floats", B, N, len(channels), ch),), dtype=torch.float32, device=dev)
        call("hp_target_forward", B, N, len(channels), ch, theta, theta.size(1), points, acts, y, current_stream(dev))
        ctx.save_for_backward(theta, points, acts)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        grad_y = grad_y.contiguous()
        theta, points = ctx.saved_tensors[:2]
        B, N = points.size(0), points.size(1)
        dev = theta.device
        grad_theta = torch.empty_like(theta)
        if ctx.fused:
            ws = torch.empty((_long_fn("hp_target_fused_workspace_floats", B, N),), dtype=torch.float32, device=dev)
            call("hp_target_fused_backward", B, N, theta, theta.size(1), points, grad_y, grad_theta, ws, current_stream(dev))
            return grad_theta, None, None
        acts = ctx.saved_tensors[2]
        ch = (c_int * len(ctx.channels))(*ctx.channels)
        ws = torch.empty((_long_fn("hp_target_backward_workspace_floats", B, N, len(ctx.channels), ch),), dtype=torch.float32,
                         device=dev)
        call("hp_target_backward", B, N, len(ctx.channels), ch, theta, theta.size(1), points,
             acts, grad_y, grad_theta, ws, current_stream(dev))
        return grad_theta, None, None


class KLDFunction(Function):
    """core/epoch_loops.py:29-30: 0.5*sum(exp(v)+mu^2-1-v)/B with v = the encoder's exp(logvar)."""

    @staticmethod
    def forward(ctx, explv, mu, batch):
        explv, mu = explv.contiguous(), mu.contiguous()
        check_input(explv, "explv")
        check_input(mu, "mu")
        out = torch.empty((), dtype=torch.float32, device=mu.device)
        call("hp_kld_forward", c_long(mu.numel()), batch, explv, mu, out, current_stream(mu.device))
        ctx.batch = batch
        ctx.save_for_backward(explv, mu)
        return out

    @staticmethod
    def backward(ctx, g):
        explv, mu = ctx.saved_tensors
        gv, gm = torch.empty_like(explv), torch.empty_like(mu)
        call("hp_kld_backward", c_long(mu.numel()), ctx.batch, explv, mu, g.contiguous(), gv, gm, current_stream(mu.device))
        return gv, gm, None


def kld_loss(explv, mu, batch=None):
    return KLDFunction.apply(explv, mu, mu.size(0) if batch is None else batch)


def sample_points(B, N, coef, seed, offset, device):
    """Decoder input points for B clouds (utils/points.py distribution) drawn on the device."""
    out = torch.empty((B, N, 3), dtype=torch.float32, device=device)
    call("hp_sample_points", c_long(B * N), float(coef), ctypes.c_ulonglong(seed & (2 ** 64 - 1)),
         ctypes.c_ulonglong(offset & (2 ** 64 - 1)), out, current_stream(device))
    return out


def slice_clouds(points, target=1024, seed=0, max_rounds=100000, planes=None):
    """Random-plane slicer (datasets/utils/dataset_generator.py:26-39) for a batch of clouds on the device.
    points (B,N,3) -> (part_with_target_points (B,target,3), rest (B,N-target,3), plane).
    planes=None: candidate planes are drawn on the device (Philox(seed)), fp32 classification; plane = (B,4) float32, the
    accepted plane.  planes = (B,R,4) or (R,4) float64 candidate (params, bias) rows — e.g. the sequence numpy's generator
    gives the reference: float64 classification exactly as HyperPlane.check_point, the reference's own split; plane = (B,)
    int32, the index of the accepted candidate."""
    points = points.contiguous()
    check_input(points, "points")
    B, N = points.size(0), points.size(1)
    dev = points.device
    a = torch.empty((B, target, 3), dtype=torch.float32, device=dev)
    b = torch.empty((B, N - target, 3), dtype=torch.float32, device=dev)
    status = torch.empty((B,), dtype=torch.int32, device=dev)
    if planes is not None:
        planes = torch.as_tensor(planes, dtype=torch.float64).to(dev)
        if planes.dim() == 2:
            planes = planes.unsqueeze(0).expand(B, -1, -1)
        if planes.dim() != 3 or planes.size(0) != B or planes.size(2) != 4 or planes.size(1) == 0:
            raise HipExtensionError("planes must be (B,R,4) or (R,4) float64 with R > 0")
        planes = planes.contiguous()
        plane = torch.empty((B,), dtype=torch.int32, device=dev)
        call("hp_slice_clouds_planes", B, N, target, points, planes, planes.size(1), a, b, plane, status, current_stream(dev))
    else:
        plane = torch.empty((B, 4), dtype=torch.float32, device=dev)
        call("hp_slice_clouds", B, N, target, points, ctypes.c_ulonglong(seed & (2 ** 64 - 1)), max_rounds, a, b, plane, status,
             current_stream(dev))
    if int(status.max().item()) != 0:     # data preparation, not the training step: a host sync is fine here
        raise HipExtensionError(f"no plane with a {target}-point side found for {int((status != 0).sum())} cloud(s)")
    return a, b, plane


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0):
    """In-place fused Adam over flat fp32 tensors (torch.optim.Adam semantics, wd=0, amsgrad=False)."""
    for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        check_input(t, n)
    call("hp_adam_step", c_long(p.numel()), p, g, m, v, float(lr), float(beta1), float(beta2), float(eps), int(step),
         float(grad_scale), current_stream(p.device))


class _GemmDesc(ctypes.Structure):
    _fields_ = [("A", c_void_p), ("B", c_void_p), ("C", c_void_p), ("bias", c_void_p), ("mask", c_void_p),
                ("add", c_void_p), ("ws", c_void_p),
                ("sAz", c_long), ("sBz", c_long), ("sCz", c_long), ("sBiasz", c_long), ("sMaskz", c_long), ("sAddz", c_long),
                ("sAi", c_long), ("sAk", c_long), ("sBk", c_long), ("sBj", c_long),
                ("ldc", c_int), ("ldmask", c_int), ("ldadd", c_int),
                ("M", c_int), ("N", c_int), ("K", c_int), ("batch", c_int), ("ksplit", c_int), ("flags", c_int),
                ("cmax", c_void_p), ("cidx", c_void_p), ("group_rows", c_int), ("rsum", c_void_p), ("sRsumz", c_long),
                ("dyn_count", c_void_p), ("dyn_kind", c_int)]


class GemmF16x2:
    """Test / bench hook over hp_gemm_f16x2_*: C = act(X W^T + b) on the f16 matrix pipe with every fp32 operand split into
    two f16 pieces (csrc/conv_split.hip — the kernel the encoders' conv stack runs).  The constructor prepares (max|X|, the
    pieces of W), `run()` is the GEMM launch alone."""

    def __init__(self, X, W, bias, relu=False, out=None):
        for t, n in ((X, "X"), (W, "W"), (bias, "bias")):
            check_input(t, n)
        self.M, self.K = X.shape
        self.N = W.shape[0]
        if W.shape[1] != self.K or bias.numel() != self.N:
            raise HipExtensionError("GemmF16x2: X (M,K), W (N,K), bias (N)")
        self.X, self.W, self.bias, self.relu = X, W, bias, int(relu)
        self.ws = torch.empty((_long_fn("hp_gemm_f16x2_workspace_floats", c_long(self.M), self.N, self.K),), dtype=torch.float32, device=X.device)
        self.C = out if out is not None else torch.empty((self.M, self.N), dtype=torch.float32, device=X.device)
        call("hp_gemm_f16x2_prepare", c_long(self.M), self.N, self.K, X, W, self.ws, current_stream(X.device))

    def run(self):
        call("hp_gemm_f16x2_run", c_long(self.M), self.N, self.K, self.X, self.bias, self.C, self.relu, self.ws,
             current_stream(self.X.device))
        return self.C


class GemmPP:
    """Test / bench hook over hp_gemm_pp_*: C = act(X W^T + b) with BOTH operands in the piece format (csrc/conv_pp.hip — the
    kernel layers 2..5 of the encoders' conv stack run since round 4): X and W are packed once by the constructor (one exponent
    per 128 rows x `xcb` channels of X, one per row of W), `run(mode)` is the matrix-core launch alone — mode 0: the layer form
    (P-format C inside the workspace; `result()` brings it to fp32), mode 1: the fused max-pool's first stage (per-128-row-tile
    column maxima of X W^T + b and their rows; `partials()`)."""

    def __init__(self, X, W, bias, relu=False, xcb=None, group_rows=None):
        for t, n in ((X, "X"), (W, "W"), (bias, "bias")):
            check_input(t, n)
        self.M, self.K = X.shape
        self.N = W.shape[0]
        self.xcb = int(xcb or min(self.K, 256))
        self.bias, self.relu = bias, int(relu)
        self.group_rows = int(group_rows or 128 * ((self.M + 127) // 128))
        self.ws = torch.empty((_long_fn("hp_gemm_pp_workspace_floats", c_long(self.M), self.N, self.K),), dtype=torch.float32,
                              device=X.device)
        self.dev = X.device
        call("hp_gemm_pp_prepare", c_long(self.M), self.N, self.K, self.xcb, X, W, self.ws, current_stream(self.dev))

    def run(self, mode=0):
        call("hp_gemm_pp_run", c_long(self.M), self.N, self.K, self.xcb, self.bias, self.relu, int(mode), self.group_rows, self.ws,
             current_stream(self.dev))

    def result(self):
        C = torch.empty((self.M, self.N), dtype=torch.float32, device=self.dev)
        call("hp_gemm_pp_unpack", c_long(self.M), self.N, self.K, self.ws, C, current_stream(self.dev))
        return C

    def partials(self):
        tiles = (self.M + 127) // 128
        cmax = torch.empty((tiles, self.N), dtype=torch.float32, device=self.dev)
        cidx = torch.empty((tiles, self.N), dtype=torch.int32, device=self.dev)
        call("hp_gemm_pp_partials", c_long(self.M), self.N, self.K, self.ws, cmax, cidx, current_stream(self.dev))
        return cmax, cidx


def gemm(A, B, bias=None, relu=False, trans_a=False, trans_b=True, mask=None, add=None, ksplit=1, rowsum=False, out=None,
         dyn_rows=None, dyn_k=None):
    """Thin test hook over hp_gemm_f32 for 2-D / 3-D (batched) fp32 tensors:
    C = epi(op(A) @ op(B)); trans_b=True means B is stored (N, K) like an nn.Linear weight."""
    batched = A.dim() == 3
    A3 = A if batched else A.unsqueeze(0)
    B3 = B if B.dim() == 3 else B.unsqueeze(0)
    for t, n in ((A3, "A"), (B3, "B")):
        check_input(t, n)
    batch = A3.size(0)
    M, K = (A3.size(2), A3.size(1)) if trans_a else (A3.size(1), A3.size(2))
    N = B3.size(1) if trans_b else B3.size(2)
    if out is not None:
        check_input(out, "out")
        C = out if out.dim() == 3 else out.unsqueeze(0)
        if tuple(C.shape) != (batch, M, N):
            raise RuntimeError(f"out must have shape {(batch, M, N)}")
    else:
        C = torch.empty((batch, M, N), dtype=torch.float32, device=A.device)
    d = _GemmDesc()
    d.A, d.B, d.C = A3.data_ptr(), B3.data_ptr(), C.data_ptr()
    d.sAz = A3.stride(0)
    d.sBz = B3.stride(0) if B3.size(0) > 1 else 0
    d.sCz = M * N
    d.sAi, d.sAk = (1, A3.size(2)) if trans_a else (A3.size(2), 1)
    d.sBk, d.sBj = (1, B3.size(2)) if trans_b else (B3.size(2), 1)
    d.ldc, d.M, d.N, d.K, d.batch, d.ksplit = N, M, N, K, batch, ksplit
    flags = 0
    if bias is not None:
        check_input(bias, "bias")
        d.bias, d.sBiasz = bias.data_ptr(), (bias.stride(0) if bias.dim() == 2 else 0)
        flags |= 1
    if relu:
        flags |= 2
    if mask is not None:
        mask3 = mask if mask.dim() == 3 else mask.unsqueeze(0)
        check_input(mask3, "mask")
        d.mask, d.sMaskz, d.ldmask = mask3.data_ptr(), M * N, N
        flags |= 4
    if add is not None:
        add3 = add if add.dim() == 3 else add.unsqueeze(0)
        check_input(add3, "add")
        d.add, d.sAddz, d.ldadd = add3.data_ptr(), M * N, N
        flags |= 8
    rs = None
    if rowsum:
        rs = torch.empty((batch, M), dtype=torch.float32, device=A.device)
        d.rsum, d.sRsumz = rs.data_ptr(), M
        flags |= 32
    d.flags = flags
    if dyn_rows is not None:      # int32 device tensor with one element: the real number of rows of A / C
        d.dyn_count, d.dyn_kind = dyn_rows.data_ptr(), 1
    if dyn_k is not None:         # ... the real contraction length
        d.dyn_count, d.dyn_kind = dyn_k.data_ptr(), 2
    ws = None
    if ksplit > 1:
        ws = torch.empty((batch * ksplit * (M * N + M),), dtype=torch.float32, device=A.device)
        d.ws = ws.data_ptr()
    call("hp_gemm_f32", ctypes.byref(d), current_stream(A.device))
    out = C if batched else C[0]
    if rowsum:
        return out, (rs if batched else rs[0])
    return out
