"""Host side of the point-transformer configuration: models/point_transformer.py:15-20 on the
HIP kernels, forward and backward.  The dense projections go through the same pnpp_fc_forward / pnpp_fc_backward path
as the heads of the other models; attention, residual LayerNorm, pooling and the input projection have their own
kernels (csrc/transformer_kernels.hip) wrapped in torch.autograd.Functions here.  No PyTorch operator computes anything.

Dropout (train mode): nn.TransformerEncoderLayer's four dropouts -- on the attention weights (bit-packed keep masks from
a counter-based generator, applied inside the attention kernels, regenerable for the backward pass), after out_proj,
after the ReLU and after linear2 (keep-masks of the projection kernels).  The random streams are this library's own:
same distribution as the reference, not the same bits.
"""
from __future__ import annotations

import torch

from . import _lib as L
from . import ops
from .ops import _f32, _scratch, _stream


class _LinearSmallK(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x2d, w, b, sinks):
        x2d, w = _f32(x2d, "x"), _f32(w, "weight")
        b = _f32(b, "bias") if b is not None else None
        M, K = x2d.shape
        N = w.shape[0]
        y = torch.empty(M, N, device=x2d.device, dtype=torch.float32)
        L.check(L.lib().pnpp_linear_smallk(x2d.data_ptr(), w.data_ptr(), None if b is None else b.data_ptr(), M, K, N,
                                           y.data_ptr(), _stream()))
        ctx.save_for_backward(x2d)
        ctx.dims, ctx.has_bias = (M, K, N), b is not None
        ctx.sinks = sinks
        return y

    @staticmethod
    def backward(ctx, dy):
        (x2d,) = ctx.saved_tensors
        M, K, N = ctx.dims
        dy = _f32(dy, "dy")
        lib = L.lib()
        sw, sb = ctx.sinks
        dw = sw if sw is not None else torch.empty(N, K, device=dy.device, dtype=torch.float32)
        db = (sb if sb is not None else torch.empty(N, device=dy.device, dtype=torch.float32)) if ctx.has_bias else None
        scratch = _scratch(lib.pnpp_linear_smallk_bwd_scratch_bytes(M, N), dy.device)
        L.check(lib.pnpp_linear_smallk_bwd(x2d.data_ptr(), dy.data_ptr(), M, K, N, dw.data_ptr(),
                                           None if db is None else db.data_ptr(), scratch.data_ptr(), _stream()))
        return None, (None if sw is not None else dw), (None if (sb is not None or db is None) else db), None


def linear_smallk(x2d: torch.Tensor, lin) -> torch.Tensor:
    """nn.Linear with at most 8 inputs (input_proj): y = x W^T + b.  The input is data: it gets no gradient."""
    # the gradient destinations are claimed here: inside Function.forward autograd is switched off and ops.grad_sink declines
    return _LinearSmallK.apply(x2d, lin.weight, lin.bias, (ops.grad_sink(lin.weight), ops.grad_sink(lin.bias)))


_att_counters = {}


def _att_counter(device: torch.device) -> torch.Tensor:
    """Device-side call counter of the attention-dropout draws: [calls so far, ticket word], bumped by the mask kernel
    itself, so a step captured in a hipGraph draws fresh masks on every replay (a host-side counter would be frozen into
    the capture and replay the same mask)."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    c = _att_counters.get(key)
    if c is None:
        c = torch.zeros(2, dtype=torch.int64, device=device)
        _att_counters[key] = c
        ops._dropout_counters[("attention",) + key] = c      # pnpp_hip.sampling.snapshot()/restore() cover it too
    return c


def attention_dropout_mask(B: int, N: int, H: int, p: float, device, seed=None, stream_id=None):
    """Bit-packed keep masks (mask, maskT), each (B,H,N,N/32) int32, for dropout p on the attention weights.  The bits are
    a pure function of (seed, stream id); seed defaults to torch.initial_seed().  With an explicit `stream_id` that is the
    id; by default it is 1 + the number of draws made so far on this device, counted in device memory."""
    if seed is None:
        seed = torch.initial_seed()
    device = torch.device(device)
    mask = torch.empty(B, H, N, N // 32, device=device, dtype=torch.int32)
    maskT = torch.empty_like(mask)
    if stream_id is None:
        L.check(L.lib().pnpp_attention_dropout_mask_dev(int(seed) & (2**64 - 1), _att_counter(device).data_ptr(), 1, B, N, H, float(p),
                                                        mask.data_ptr(), maskT.data_ptr(), _stream()))
    else:
        L.check(L.lib().pnpp_attention_dropout_mask(int(seed) & (2**64 - 1), int(stream_id) & (2**64 - 1), B, N, H, float(p),
                                                    mask.data_ptr(), maskT.data_ptr(), _stream()))
    return mask, maskT


class _Attention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, num_heads, p, masks, n_valid):
        qkv = _f32(qkv, "qkv")
        B, N, E3 = qkv.shape
        if N % 128 != 0 or not 0 < n_valid <= N:
            raise ValueError(f"attention: {N} rows per cloud (must be a multiple of 128: pad) with {n_valid} valid points")
        if E3 % (3 * num_heads) != 0:
            raise ValueError(f"attention: last dimension {E3} is not 3 * heads * head_dim")
        E = E3 // 3
        mask, maskT = masks if masks is not None else (None, None)
        out = torch.empty(B, N, E, device=qkv.device, dtype=torch.float32)
        lse = torch.empty(B, num_heads, N, device=qkv.device, dtype=torch.float32)
        L.check(L.lib().pnpp_attention_fwd(qkv.data_ptr(), B, N, int(n_valid), num_heads, E // num_heads,
                                           None if mask is None else mask.data_ptr(), float(p), out.data_ptr(), lse.data_ptr(),
                                           _stream()))
        ctx.save_for_backward(qkv, out, lse)
        ctx.heads, ctx.p, ctx.masks, ctx.n_valid = num_heads, float(p), (mask, maskT), int(n_valid)
        ctx.mark_non_differentiable(lse)
        return out, lse

    @staticmethod
    def backward(ctx, d_out, _d_lse):
        qkv, out, lse = ctx.saved_tensors
        B, N, E3 = qkv.shape
        d_out = _f32(d_out, "d_out")
        dqkv = torch.empty_like(qkv)
        dsum = torch.empty_like(lse)
        mask, maskT = ctx.masks
        L.check(L.lib().pnpp_attention_bwd(qkv.data_ptr(), out.data_ptr(), d_out.data_ptr(), lse.data_ptr(), B, N, ctx.n_valid, ctx.heads,
                                           E3 // 3 // ctx.heads, None if mask is None else mask.data_ptr(),
                                           None if maskT is None else maskT.data_ptr(), ctx.p, dqkv.data_ptr(), dsum.data_ptr(),
                                           _stream()))
        return dqkv, None, None, None, None


def attention(qkv: torch.Tensor, num_heads: int, want_lse: bool = False, p: float = 0.0, masks=None, n_valid=None):
    """qkv (B,N,3E), in_proj bias included -> (B,N,E) [, log-sum-exp of the scaled scores (B,H,N)].
    N (rows per cloud) must be a multiple of 128; n_valid <= N (default N) says how many of them are points -- the rest is
    the caller's padding: it receives no attention weight, its own output rows are to be ignored and its d_out must be zero.
    p > 0 applies dropout to the attention weights with the keep bits `masks` = attention_dropout_mask(...)
    (drawn here when not given)."""
    if p > 0.0 and masks is None:
        masks = attention_dropout_mask(qkv.shape[0], qkv.shape[1], num_heads, p, qkv.device)
    out, lse = _Attention.apply(qkv, num_heads, p if masks is not None else 0.0, masks, qkv.shape[1] if n_valid is None else n_valid)
    return (out, lse) if want_lse else out


class _AddLayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x2d, r2d, w, b, eps, sinks):
        x2d, w, b = _f32(x2d, "x"), _f32(w, "weight"), _f32(b, "bias")
        r2d = _f32(r2d, "r") if r2d is not None else None
        M, E = x2d.shape
        y = torch.empty_like(x2d)
        L.check(L.lib().pnpp_add_layernorm(x2d.data_ptr(), None if r2d is None else r2d.data_ptr(), w.data_ptr(), b.data_ptr(),
                                           M, E, float(eps), y.data_ptr(), _stream()))
        ctx.save_for_backward(x2d, r2d if r2d is not None else x2d.new_empty(0), w)
        ctx.has_r, ctx.eps = r2d is not None, float(eps)
        ctx.sinks = sinks
        return y

    @staticmethod
    def backward(ctx, dy):
        x2d, r2d, w = ctx.saved_tensors
        r2d = r2d if ctx.has_r else None
        M, E = x2d.shape
        dy = _f32(dy, "dy")
        lib = L.lib()
        du = torch.empty_like(x2d)
        dwb = torch.empty(2, E, device=dy.device, dtype=torch.float32)
        scratch = _scratch(lib.pnpp_add_layernorm_bwd_scratch_bytes(M, E), dy.device)
        L.check(lib.pnpp_add_layernorm_bwd(x2d.data_ptr(), None if r2d is None else r2d.data_ptr(), w.data_ptr(), dy.data_ptr(),
                                           M, E, ctx.eps, du.data_ptr(), dwb.data_ptr(), scratch.data_ptr(), _stream()))
        sw, sb = ctx.sinks
        if sw is not None:
            sw.copy_(dwb[0])
        if sb is not None:
            sb.copy_(dwb[1])
        return du, (du if ctx.has_r else None), (None if sw is not None else dwb[0]), (None if sb is not None else dwb[1]), None, None


def add_layernorm(x2d: torch.Tensor, r2d, norm: torch.nn.LayerNorm) -> torch.Tensor:
    """LayerNorm(x + r) over the last dimension (the post-norm residual blocks of nn.TransformerEncoderLayer)."""
    return _AddLayerNorm.apply(x2d, r2d, norm.weight, norm.bias, norm.eps, (ops.grad_sink(norm.weight), ops.grad_sink(norm.bias)))


class _MeanPoints(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _f32(x, "x")
        B, N, E = x.shape
        y = torch.empty(B, E, device=x.device, dtype=torch.float32)
        L.check(L.lib().pnpp_mean_points(x.data_ptr(), B, N, E, y.data_ptr(), _stream()))
        ctx.dims = (B, N, E)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, N, E = ctx.dims
        dy = _f32(dy, "dy")
        dx = torch.empty(B, N, E, device=dy.device, dtype=torch.float32)
        L.check(L.lib().pnpp_mean_points_bwd(dy.data_ptr(), B, N, E, dx.data_ptr(), _stream()))
        return dx


def mean_points(x: torch.Tensor) -> torch.Tensor:
    return _MeanPoints.apply(x)


class _Affine:
    """A (weight, bias) pair presented like nn.Linear to ops.fc_block (MultiheadAttention keeps in_proj as raw tensors)."""

    def __init__(self, weight, bias):
        self.weight, self.bias = weight, bias


def point_transformer_forward(model, xyz: torch.Tensor) -> torch.Tensor:
    """models/point_transformer.py:15-20, differentiable.  In train mode the four dropouts of every
    nn.TransformerEncoderLayer are applied: on the attention weights inside the attention kernels (counter-based keep
    bits), after out_proj (dropout1), after the ReLU (dropout) and after linear2 (dropout2) as keep-masks of the
    projection kernels."""
    xyz = _f32(xyz, "xyz")
    B, n_pts, K = xyz.shape
    N = (n_pts + 127) // 128 * 128          # rows per cloud: the attention kernels walk 128-query / 32-key blocks
    if N != n_pts:                          # any cloud size (the reference's data has 10,000 points): zero rows are appended,
        pad = xyz.new_zeros(B, N, K)        # the attention kernels give them no weight, and the pooling below leaves them out
        pad[:, :n_pts] = xyz
        xyz = pad
    x = linear_smallk(xyz.reshape(B * N, K), model.input_proj)                        # (B*N, E)
    E = x.shape[1]
    tr = model.training
    for layer in model.transformer.layers:
        att = layer.self_attn
        if layer.norm_first or att.in_proj_weight is None or not att.batch_first:
            raise NotImplementedError("only the post-norm, packed in_proj, batch_first encoder layer of the reference")
        qkv = ops.fc_block(x, _Affine(att.in_proj_weight, att.in_proj_bias), training=tr)   # (B*N, 3E), bias added
        o = attention(qkv.view(B, N, 3 * E), att.num_heads, p=att.dropout if tr else 0.0, n_valid=n_pts).view(B * N, E)
        o = ops.fc_block(o, att.out_proj, dropout=layer.dropout1, training=tr)
        x = add_layernorm(x, o, layer.norm1)
        hid = ops.fc_block(x, layer.linear1, relu=True, dropout=layer.dropout, training=tr)   # dropout(relu(W1 x + b1))
        f = ops.fc_block(hid, layer.linear2, dropout=layer.dropout2, training=tr)
        x = add_layernorm(x, f, layer.norm2)
    pooled = mean_points(x.view(B, N, E) if N == n_pts else x.view(B, N, E)[:, :n_pts])
    return ops.fc_block(pooled, model.fc_out, training=tr)
