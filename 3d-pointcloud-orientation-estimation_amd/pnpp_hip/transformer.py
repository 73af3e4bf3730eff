"""Host side of the point-transformer configuration: models/point_transformer.py:15-20 on the
HIP kernels.  Forward only (see models/point_transformer.py for the status): the dense projections go through the
same pnpp_fc_forward path as the heads of the other models, attention / residual LayerNorm / pooling have their own
kernels (csrc/transformer_kernels.hip).  Everything runs under torch.no_grad(); no PyTorch operator computes anything.
"""
from __future__ import annotations

import ctypes as C  # noqa: F401

import torch

from . import _lib as L
from . import ops
from .ops import _f32, _stream


def linear_smallk(x2d: torch.Tensor, lin: torch.nn.Linear) -> torch.Tensor:
    x2d = _f32(x2d, "x")
    w, b = _f32(lin.weight, "weight"), (_f32(lin.bias, "bias") if lin.bias is not None else None)
    M, K = x2d.shape
    N = w.shape[0]
    y = torch.empty(M, N, device=x2d.device, dtype=torch.float32)
    L.check(L.lib().pnpp_linear_smallk(x2d.data_ptr(), w.data_ptr(), None if b is None else b.data_ptr(), M, K, N, y.data_ptr(),
                                       _stream()))
    return y


def attention(qkv: torch.Tensor, num_heads: int, want_lse: bool = False):
    """qkv (B,N,3E) -> (B,N,E) [, lse (B,H,N)]."""
    qkv = _f32(qkv, "qkv")
    B, N, E3 = qkv.shape
    if E3 % (3 * num_heads) != 0:
        raise ValueError(f"attention: last dimension {E3} is not 3 * heads * head_dim")
    E = E3 // 3
    out = torch.empty(B, N, E, device=qkv.device, dtype=torch.float32)
    lse = torch.empty(B, num_heads, N, device=qkv.device, dtype=torch.float32) if want_lse else None
    L.check(L.lib().pnpp_attention_fwd(qkv.data_ptr(), B, N, num_heads, E // num_heads, out.data_ptr(),
                                       None if lse is None else lse.data_ptr(), _stream()))
    return (out, lse) if want_lse else out


def add_layernorm(x2d: torch.Tensor, r2d, norm: torch.nn.LayerNorm) -> torch.Tensor:
    x2d = _f32(x2d, "x")
    r2d = _f32(r2d, "r") if r2d is not None else None
    M, E = x2d.shape
    y = torch.empty_like(x2d)
    L.check(L.lib().pnpp_add_layernorm(x2d.data_ptr(), None if r2d is None else r2d.data_ptr(), _f32(norm.weight, "w").data_ptr(),
                                       _f32(norm.bias, "b").data_ptr(), M, E, float(norm.eps), y.data_ptr(), _stream()))
    return y


def mean_points(x: torch.Tensor) -> torch.Tensor:
    x = _f32(x, "x")
    B, N, E = x.shape
    y = torch.empty(B, E, device=x.device, dtype=torch.float32)
    L.check(L.lib().pnpp_mean_points(x.data_ptr(), B, N, E, y.data_ptr(), _stream()))
    return y


@torch.no_grad()
def point_transformer_forward(model, xyz: torch.Tensor) -> torch.Tensor:
    """models/point_transformer.py:15-20 with the dropouts inactive (eval)."""
    xyz = _f32(xyz, "xyz")
    B, N, K = xyz.shape
    x = linear_smallk(xyz.reshape(B * N, K), model.input_proj)                        # (B*N, E)
    E = x.shape[1]
    for layer in model.transformer.layers:
        att = layer.self_attn
        if layer.norm_first or att.in_proj_weight is None or not att.batch_first:
            raise NotImplementedError("only the post-norm, packed in_proj, batch_first encoder layer of the reference")
        in_proj = _Affine(att.in_proj_weight, att.in_proj_bias)
        qkv = ops.fc_block(x, in_proj, training=False)                                  # (B*N, 3E), bias added
        o = attention(qkv.view(B, N, 3 * E), att.num_heads).view(B * N, E)
        o = ops.fc_block(o, att.out_proj, training=False)
        x = add_layernorm(x, o, layer.norm1)
        hid = ops.fc_block(x, layer.linear1, relu=True, training=False)                 # relu(W1 x + b1)
        f = ops.fc_block(hid, layer.linear2, training=False)
        x = add_layernorm(x, f, layer.norm2)
    pooled = mean_points(x.view(B, N, E))
    return ops.fc_block(pooled, model.fc_out, training=False)


class _Affine:
    """A (weight, bias) pair presented like nn.Linear to ops.fc_block (MultiheadAttention keeps in_proj as raw tensors)."""

    def __init__(self, weight, bias):
        self.weight, self.bias = weight, bias
