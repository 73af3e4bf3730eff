"""Host side of the HIP operators: argument checking, workspace allocation (torch caching
allocator), stream plumbing and torch.autograd glue around the C ABI of libpnpp_hip.so.

Every function here requires float32 tensors on an AMD GPU; there is no CPU implementation.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib as L


# ------------------------------------------------------------------------------------------------
# plumbing
# ------------------------------------------------------------------------------------------------
def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(t: torch.Tensor, name: str) -> None:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} is on '{t.device}': the pnpp HIP operators run on an AMD GPU only "
                           "(no CPU fallback exists in this package)")


def _f32(t: torch.Tensor, name: str) -> torch.Tensor:
    _need_gpu(t, name)
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    return t.contiguous()


def _i32(t: torch.Tensor, name: str) -> torch.Tensor:
    _need_gpu(t, name)
    if t.dtype not in (torch.int32, torch.int64):
        raise TypeError(f"{name} must be an integer tensor, got {t.dtype}")
    return t.to(torch.int32).contiguous()


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _nbt(bn) -> Optional[torch.Tensor]:
    """A BatchNorm's num_batches_tracked as the int64 device scalar the kernels increment (None when it is not tracked)."""
    t = getattr(bn, "num_batches_tracked", None)
    if t is None:
        return None
    _need_gpu(t, "num_batches_tracked")
    if t.dtype != torch.int64:
        raise TypeError(f"num_batches_tracked must be int64, got {t.dtype}")
    return t


def grad_sink(p: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """The slice of an optimiser's flat gradient buffer that the backward kernels may OVERWRITE with p's gradient
    (FlatAdam(fused_grads=True)), or None when the gradient has to go through autograd's accumulation instead.

    Overwriting is only right for the first use of a parameter between two zero_grad()/step() calls.  A second use --
    gradient accumulation over micro-batches, two forward passes summed into one loss, shared weights -- gets None: its
    gradient is then returned to autograd, which ADDS it to p.grad (the same memory), so nothing is lost.  The claim is
    taken when the forward pass records the destination; FlatAdam.zero_grad()/step() release it."""
    if p is None or not torch.is_grad_enabled():
        return None
    sink = getattr(p, "_pnpp_grad_sink", None)
    if sink is None or getattr(p, "_pnpp_sink_claimed", False):
        return None
    p._pnpp_sink_claimed = True
    return sink


_dropout_counters = {}


def _dropout_counter(device: torch.device) -> torch.Tensor:
    """Device-side stream id of the in-kernel dropout draws: [counter, ticket word], bumped by the kernels themselves."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    c = _dropout_counters.get(key)
    if c is None:
        c = torch.zeros(2, dtype=torch.int64, device=device)
        _dropout_counters[key] = c
    return c


_scratch_cache = {}


def _scratch(nbytes: int, device: torch.device) -> torch.Tensor:
    """Transient workspace, reused per (device, stream): all consumers are ordered on that stream."""
    key = (device.index, _stream())
    buf = _scratch_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _scratch_cache[key] = buf
    return buf


def set_matmul_precision(mode: str) -> None:
    """'f32' (default: exact float32 MFMA, the reference's arithmetic) or 'bf16' (throughput mode: the grouped layers' large
    GEMMs round their two operands to bfloat16 and accumulate in float32; everything else stays float32 / float64).
    Process-wide; a hipGraph captured under one mode keeps it."""
    if mode not in ("f32", "bf16"):
        raise ValueError(f"matmul precision must be 'f32' or 'bf16', got {mode!r}")
    L.check(L.lib().pnpp_set_matmul_precision(1 if mode == "bf16" else 0))


def get_matmul_precision() -> str:
    return "bf16" if L.lib().pnpp_get_matmul_precision() else "f32"


def set_float32_products(mode: str) -> None:
    """How the float32 products of the grouped layers' large GEMMs are formed: 'split' (default: six exact bf16 x bf16 partial products
    of three-way operand splits on the bf16 matrix pipe, float32 accumulation -- float32 results to float32 rounding, 2.67 x the
    float32 MFMA rate; csrc/gemm_wsf3_kernels.hip) or 'mfma' (v_mfma_f32_32x32x2_f32).  Process-wide; unrelated to
    set_matmul_precision, which ROUNDS operands to one bfloat16."""
    if mode not in ("split", "mfma"):
        raise ValueError(f"float32 products must be 'split' or 'mfma', got {mode!r}")
    L.check(L.lib().pnpp_set_split_products(1 if mode == "split" else 0))


def get_float32_products() -> str:
    return "split" if L.lib().pnpp_get_split_products() else "mfma"


# ------------------------------------------------------------------------------------------------
# index primitives
# ------------------------------------------------------------------------------------------------
def square_distance(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    src, dst = _f32(src, "src"), _f32(dst, "dst")
    B, S, _ = src.shape
    N = dst.shape[1]
    out = torch.empty(B, S, N, device=src.device, dtype=torch.float32)
    L.check(L.lib().pnpp_square_distance(src.data_ptr(), dst.data_ptr(), B, S, N, out.data_ptr(), _stream()))
    return out


def knn(new_xyz: torch.Tensor, xyz: torch.Tensor, k: int) -> torch.Tensor:
    """(B,S,k) int32 neighbour indices, ascending (distance, index)."""
    new_xyz, xyz = _f32(new_xyz, "new_xyz"), _f32(xyz, "xyz")
    B, S, _ = new_xyz.shape
    N = xyz.shape[1]
    idx = torch.empty(B, S, k, device=xyz.device, dtype=torch.int32)
    L.check(L.lib().pnpp_knn(new_xyz.data_ptr(), xyz.data_ptr(), B, S, N, int(k), idx.data_ptr(), _stream()))
    return idx


def farthest_point_sample(xyz: torch.Tensor, npoint: int, start: Optional[torch.Tensor] = None) -> torch.Tensor:
    xyz = _f32(xyz, "xyz")
    B, N, _ = xyz.shape
    if start is None:  # PointNet++Demo.py:20 draws the first index with torch.randint
        start = torch.randint(0, N, (B,), dtype=torch.long)
    start = start.to(device=xyz.device, dtype=torch.int32).contiguous()
    out = torch.empty(B, npoint, device=xyz.device, dtype=torch.int32)
    L.check(L.lib().pnpp_fps(xyz.data_ptr(), B, N, int(npoint), start.data_ptr(), out.data_ptr(), _stream()))
    return out


def ball_query(radius: float, nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor) -> torch.Tensor:
    xyz, new_xyz = _f32(xyz, "xyz"), _f32(new_xyz, "new_xyz")
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    idx = torch.empty(B, S, nsample, device=xyz.device, dtype=torch.int32)
    L.check(L.lib().pnpp_ball_query(new_xyz.data_ptr(), xyz.data_ptr(), B, S, N, float(radius), int(nsample),
                                    idx.data_ptr(), _stream()))
    return idx


def sample_random(seed: int, stream_id: int, B: int, N: int, npoint: int, device) -> torch.Tensor:
    out = torch.empty(B, npoint, device=device, dtype=torch.int32)
    L.check(L.lib().pnpp_sample_random(int(seed) & (2**64 - 1), int(stream_id) & (2**64 - 1), B, N, int(npoint),
                                       out.data_ptr(), _stream()))
    return out


def sample_random_dev(seed: int, counter: torch.Tensor, offset: int, B: int, N: int, npoint: int) -> torch.Tensor:
    """Like sample_random, with stream id = counter[0] + offset read at kernel time; the kernel then adds 1 to
    counter[0].  `counter` is an int64 device tensor of two words (call counter, ticket word kept at zero)."""
    _need_gpu(counter, "counter")
    if counter.dtype != torch.int64 or counter.numel() != 2 or not counter.is_contiguous():
        raise ValueError("counter must be a contiguous int64 tensor of 2 elements")
    out = torch.empty(B, npoint, device=counter.device, dtype=torch.int32)
    L.check(L.lib().pnpp_sample_random_dev(int(seed) & (2**64 - 1), counter.data_ptr(), int(offset) & (2**64 - 1), B, N,
                                           int(npoint), out.data_ptr(), _stream()))
    return out


def sample_random_dev2(seed: int, counter: torch.Tensor, offset: int, B: int, N1: int, npoint1: int, N2: int, npoint2: int):
    """Two consecutive sample_random_dev draws in one launch: (B,npoint1) from N1 with the counter's value, (B,npoint2)
    from N2 with the next one; counter[0] += 2.  Bit-identical to the two separate calls."""
    _need_gpu(counter, "counter")
    if counter.dtype != torch.int64 or counter.numel() != 2 or not counter.is_contiguous():
        raise ValueError("counter must be a contiguous int64 tensor of 2 elements")
    out1 = torch.empty(B, npoint1, device=counter.device, dtype=torch.int32)
    out2 = torch.empty(B, npoint2, device=counter.device, dtype=torch.int32)
    L.check(L.lib().pnpp_sample_random_dev2(int(seed) & (2**64 - 1), counter.data_ptr(), int(offset) & (2**64 - 1), B, N1,
                                            int(npoint1), out1.data_ptr(), N2, int(npoint2), out2.data_ptr(), _stream()))
    return out1, out2


def subsample_points(seed: int, stream_id: int, bank: torch.Tensor, lengths: torch.Tensor, num: int,
                     cloud_ids: Optional[torch.Tensor] = None) -> torch.Tensor:
    """`num` points of each selected cloud of a device-resident bank (n_clouds,Lmax,3): without replacement where the
    cloud has at least `num` points, with replacement otherwise (dataloader_*: sample_pts).  Returns (B,num,3)."""
    bank = _f32(bank, "bank")
    _need_gpu(lengths, "lengths")
    if bank.dim() != 3 or bank.shape[2] != 3 or lengths.dtype != torch.int32 or lengths.numel() != bank.shape[0]:
        raise ValueError("subsample_points: bank must be (n_clouds,Lmax,3) float32 and lengths (n_clouds,) int32")
    lengths = lengths.contiguous()
    if cloud_ids is not None:
        _need_gpu(cloud_ids, "cloud_ids")
        cloud_ids = cloud_ids.to(torch.int32).contiguous()
        if cloud_ids.numel() and (int(cloud_ids.min()) < 0 or int(cloud_ids.max()) >= bank.shape[0]):
            raise IndexError("subsample_points: cloud id out of range")
    B = bank.shape[0] if cloud_ids is None else cloud_ids.numel()
    out = torch.empty(B, int(num), 3, device=bank.device, dtype=torch.float32)
    L.check(L.lib().pnpp_subsample_points(int(seed) & (2**64 - 1), int(stream_id) & (2**64 - 1), bank.data_ptr(),
                                          lengths.data_ptr(), _p(cloud_ids), B, bank.shape[1], int(num), out.data_ptr(),
                                          _stream()))
    return out


class _IndexPoints(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx):
        points = _f32(points, "points")
        idx32 = _i32(idx, "idx")
        B, N, Cc = points.shape
        M = idx32[0].numel()
        out = torch.empty(*idx32.shape, Cc, device=points.device, dtype=torch.float32)
        L.check(L.lib().pnpp_index_points(points.data_ptr(), idx32.data_ptr(), B, N, Cc, M, out.data_ptr(), _stream()))
        ctx.save_for_backward(idx32)
        ctx.shape = (B, N, Cc, M)
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx32,) = ctx.saved_tensors
        B, N, Cc, M = ctx.shape
        dout = _f32(dout, "dout")
        dpoints = torch.zeros(B, N, Cc, device=dout.device, dtype=torch.float32)
        L.check(L.lib().pnpp_index_points_bwd(dout.data_ptr(), idx32.data_ptr(), B, N, Cc, M, dpoints.data_ptr(), _stream()))
        return dpoints, None


def index_points(points: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    return _IndexPoints.apply(points, idx)


# ------------------------------------------------------------------------------------------------
# set abstraction
# ------------------------------------------------------------------------------------------------
def _sa_desc(B, N, S, K, D, channels: Sequence[int], group_all: bool, training: bool, eps: float, momentum: float):
    d = L.SaDesc()
    d.B, d.N, d.S, d.K, d.D, d.L = B, N, S, K, D, len(channels)
    for i, c in enumerate(channels):
        d.C[i] = int(c)
    d.group_all, d.training = int(group_all), int(training)
    d.eps, d.momentum = float(eps), float(momentum)
    return d


def _ptr_array(tensors: Sequence[Optional[torch.Tensor]]):
    arr = (C.c_void_p * L.PNPP_MAX_LAYERS)()
    for i, t in enumerate(tensors):
        arr[i] = _p(t)
    return arr


# set to a list to receive, per set-abstraction forward call, {"neighbours": (B,S,K) int32 | None, "argmax": (B,S,C) int32}
# (views into the call's saved workspace) -- used by the parity tests to hand the fp64 oracle the same max-pool routing
sa_tap: Optional[list] = None


class _SetAbstraction(torch.autograd.Function):
    """PointNetSetAbstraction.forward / backward as two C calls (models/pointnet_pp_8dir.py:21-43)."""

    @staticmethod
    def forward(ctx, xyz, points, centre_idx, neighbour_idx, cfg, running, *params):
        # params: L x (conv_w, conv_b, bn_w, bn_b); running: L x (running_mean, running_var)
        K, group_all, training, eps, momentum, sinks, nbt, pre = cfg
        Lh = len(params) // 4
        xyz = _f32(xyz, "xyz")
        B, N, _ = xyz.shape
        D = 0
        if points is not None:
            points = _f32(points, "points")
            D = points.shape[2]
        conv_w = [_f32(params[4 * l], "conv weight") for l in range(Lh)]
        conv_b = [_f32(params[4 * l + 1], "conv bias") for l in range(Lh)]
        bn_w = [_f32(params[4 * l + 2], "bn weight") for l in range(Lh)]
        bn_b = [_f32(params[4 * l + 3], "bn bias") for l in range(Lh)]
        channels = [w.shape[0] for w in conv_w]
        if group_all:
            S, Kk = 1, N
        else:
            centre_idx = _i32(centre_idx, "centre_idx")
            S, Kk = centre_idx.shape[1], int(K)
            if neighbour_idx is not None:
                neighbour_idx = _i32(neighbour_idx, "neighbour_idx")
        desc = _sa_desc(B, N, S, Kk, D, channels, group_all, training, eps, momentum)
        lib = L.lib()
        sb = lib.pnpp_sa_saved_bytes(C.byref(desc))
        if sb == 0:
            L.check(L.PNPP_ERR_ARG if L.last_error() else L.PNPP_ERR_ARG)
        if pre is not None:   # grouped ahead of time (group_pair): neighbours and centres already sit in this call's workspace
            saved, new_xyz = pre["saved"], pre["new_xyz"]
            if saved.numel() != sb or tuple(new_xyz.shape) != (B, S, 3) or neighbour_idx is not None:
                raise ValueError("set_abstraction: the pre-grouped workspace does not belong to this call")
        else:
            saved = torch.empty(sb, dtype=torch.uint8, device=xyz.device)
            new_xyz = torch.empty(B, S, 3, device=xyz.device, dtype=torch.float32)
        scratch = _scratch(lib.pnpp_sa_scratch_bytes(C.byref(desc)), xyz.device)
        out = torch.empty(B, S, channels[-1], device=xyz.device, dtype=torch.float32)
        a = L.SaFwdArgs()
        a.xyz, a.points = xyz.data_ptr(), _p(points)
        a.centre_idx = None if group_all else centre_idx.data_ptr()
        a.neighbour_idx = _p(neighbour_idx)
        if pre is not None:   # "already in place": the neighbour argument IS the workspace's own index block
            a.neighbour_idx = lib.pnpp_sa_saved_neighbours(C.byref(desc), saved.data_ptr())
        a.conv_w, a.conv_b, a.bn_w, a.bn_b = _ptr_array(conv_w), _ptr_array(conv_b), _ptr_array(bn_w), _ptr_array(bn_b)
        a.bn_rm = _ptr_array([running[2 * l] for l in range(Lh)])
        a.bn_rv = _ptr_array([running[2 * l + 1] for l in range(Lh)])
        a.bn_nbt = _ptr_array(nbt if nbt is not None else [None] * Lh)
        a.new_xyz, a.out, a.saved, a.scratch = new_xyz.data_ptr(), out.data_ptr(), saved.data_ptr(), scratch.data_ptr()
        L.check(lib.pnpp_sa_forward(C.byref(desc), C.byref(a), _stream()))
        ctx.desc = desc
        ctx.sinks = sinks
        ctx.has_points = points is not None
        # which kernels a level takes depends on process-wide switches (bf16 operands, SyncBN); a level on raw coordinates keeps no
        # Z_0 for the generic backward kernels, so its backward pass must run under the switches its forward pass ran under
        ctx.path_state = (lib.pnpp_get_matmul_precision(), lib.pnpp_stats_exchange_enabled(), lib.pnpp_get_split_products())
        ctx.save_for_backward(xyz, points if points is not None else xyz.new_empty(0), saved, *conv_w, *bn_w, *bn_b)
        if group_all:
            nbr = torch.empty(0, dtype=torch.int32, device=xyz.device)
        else:  # view of the neighbour indices kept in the saved workspace
            off = lib.pnpp_sa_saved_neighbours(C.byref(desc), saved.data_ptr()) - saved.data_ptr()
            nbr = saved[off:off + 4 * B * S * Kk].view(torch.int32).view(B, S, Kk)
        if sa_tap is not None:   # diagnostics: views of what backward will route by (neighbour rows, max-pool positions)
            aoff = lib.pnpp_sa_saved_argmax(C.byref(desc), saved.data_ptr()) - saved.data_ptr()
            arg = saved[aoff:aoff + 4 * B * S * channels[-1]].view(torch.int32).view(B, S, channels[-1])
            # ... and of every ReLU decision the backward pass will take (pnpp_sa_saved_relu_mask), per layer (B, S, K, C_l) uint8
            masks = []
            for l in range(Lh):
                m = torch.empty(B, S, Kk, channels[l], device=xyz.device, dtype=torch.uint8)
                L.check(lib.pnpp_sa_saved_relu_mask(C.byref(desc), saved.data_ptr(), xyz.data_ptr(), conv_w[0].data_ptr(), l,
                                                    m.data_ptr(), _stream()))
                masks.append(m)
            sa_tap.append({"neighbours": None if group_all else nbr, "argmax": arg, "relu_masks": masks})
        ctx.mark_non_differentiable(new_xyz, nbr)
        ctx.set_materialize_grads(False)  # no zero tensors (= fill launches) for the two outputs that carry no gradient
        return new_xyz, out, nbr

    @staticmethod
    def backward(ctx, _dnew_xyz, dout, _dnbr):
        desc = ctx.desc
        Lh = desc.L
        if dout is None:
            return (None,) * (6 + 4 * Lh)
        t = ctx.saved_tensors
        xyz, points, saved = t[0], (t[1] if ctx.has_points else None), t[2]
        conv_w, bn_w, bn_b = t[3:3 + Lh], t[3 + Lh:3 + 2 * Lh], t[3 + 2 * Lh:3 + 3 * Lh]
        dout = _f32(dout, "dout")
        lib = L.lib()
        if ctx.path_state != (lib.pnpp_get_matmul_precision(), lib.pnpp_stats_exchange_enabled(), lib.pnpp_get_split_products()):
            raise ValueError("set_abstraction: matmul precision or SyncBN was switched between this level's forward pass and its "
                             "backward pass; the kept workspace belongs to the kernels the forward pass took")
        scratch = _scratch(lib.pnpp_sa_scratch_bytes(C.byref(desc)), xyz.device)
        # gradient destinations: a parameter's flat-buffer sink when an optimiser registered one (the kernels then
        # write straight into the flat gradient buffer and autograd has nothing to accumulate), else fresh tensors
        sinks = ctx.sinks or [None] * (4 * Lh)
        d_conv_w = [sinks[4 * l] if sinks[4 * l] is not None else torch.empty_like(conv_w[l]) for l in range(Lh)]
        d_conv_b = [sinks[4 * l + 1] if sinks[4 * l + 1] is not None else
                    torch.empty(conv_w[l].shape[0], device=xyz.device, dtype=torch.float32) for l in range(Lh)]
        d_bn_w = [sinks[4 * l + 2] if sinks[4 * l + 2] is not None else torch.empty_like(bn_w[l]) for l in range(Lh)]
        d_bn_b = [sinks[4 * l + 3] if sinks[4 * l + 3] is not None else torch.empty_like(bn_b[l]) for l in range(Lh)]
        dpoints = torch.empty_like(points) if (points is not None and ctx.needs_input_grad[1]) else None
        a = L.SaBwdArgs()
        a.xyz, a.points = xyz.data_ptr(), _p(points)
        a.conv_w, a.bn_w, a.bn_b = _ptr_array(conv_w), _ptr_array(bn_w), _ptr_array(bn_b)
        a.dout, a.saved, a.scratch = dout.data_ptr(), saved.data_ptr(), scratch.data_ptr()
        a.d_conv_w, a.d_conv_b = _ptr_array(d_conv_w), _ptr_array(d_conv_b)
        a.d_bn_w, a.d_bn_b = _ptr_array(d_bn_w), _ptr_array(d_bn_b)
        a.dpoints = _p(dpoints)
        L.check(lib.pnpp_sa_backward(C.byref(desc), C.byref(a), _stream()))
        grads: List[Optional[torch.Tensor]] = []
        for l in range(Lh):
            for j, t in enumerate((d_conv_w[l], d_conv_b[l], d_bn_w[l], d_bn_b[l])):
                grads.append(None if sinks[4 * l + j] is not None else t)
        return (None, dpoints, None, None, None, None, *grads)


def group_pair(xyz, centre1, centre2, nsample1, convs1, nsample2, convs2, training=True):
    """Centre gather + neighbour search of two stacked set-abstraction levels (sa1 on the cloud, sa2 on sa1's centres) in ONE
    launch, ahead of both forward passes (models/pointnet_pp_8dir.py:28-31 of each level): level 2 searches among level 1's
    centres, which are rows of the same cloud, so it does not wait for level 1's MLP.  centre1 (B,S1) rows of xyz, centre2
    (B,S2) positions among level 1's centres.  Returns the two `pre=` workspaces for set_abstraction; the neighbour sets are
    those the per-level search finds (same coordinates, same arithmetic)."""
    xyz = _f32(xyz, "xyz")
    B, N, _ = xyz.shape
    c1, c2 = _i32(centre1, "centre_idx"), _i32(centre2, "centre_idx")
    S1, S2 = c1.shape[1], c2.shape[1]
    ch1, ch2 = [c.weight.shape[0] for c in convs1], [c.weight.shape[0] for c in convs2]
    d1 = _sa_desc(B, N, S1, int(nsample1), 0, ch1, False, training, 1e-5, 0.1)
    d2 = _sa_desc(B, S1, S2, int(nsample2), ch1[-1], ch2, False, training, 1e-5, 0.1)
    lib = L.lib()
    pres = []
    for d, S in ((d1, S1), (d2, S2)):
        sb = lib.pnpp_sa_saved_bytes(C.byref(d))
        if sb == 0:
            L.check(L.PNPP_ERR_ARG)
        pres.append({"saved": torch.empty(sb, dtype=torch.uint8, device=xyz.device),
                     "new_xyz": torch.empty(B, S, 3, device=xyz.device, dtype=torch.float32)})
    L.check(lib.pnpp_sa_group_pair(C.byref(d1), C.byref(d2), xyz.data_ptr(), c1.data_ptr(), c2.data_ptr(), pres[0]["saved"].data_ptr(),
                                   pres[0]["new_xyz"].data_ptr(), pres[1]["saved"].data_ptr(), pres[1]["new_xyz"].data_ptr(), _stream()))
    return pres[0], pres[1]


def set_abstraction(xyz, points, centre_idx, nsample, group_all, training, convs, bns, neighbour_idx=None,
                    return_neighbours=False, pre=None):
    """Functional form used by models.pointnet_pp_8dir.PointNetSetAbstraction.

    convs / bns are the nn.Conv2d / nn.BatchNorm2d containers (their tensors are used in place:
    weights are read, running statistics are updated by the kernels when training).
    Returns (new_xyz, new_points) exactly like the reference module.
    """
    params, running = [], []
    for conv, bn in zip(convs, bns):
        params += [conv.weight, conv.bias, bn.weight, bn.bias]
        running += [bn.running_mean, bn.running_var]
    eps = bns[0].eps
    momentum = bns[0].momentum if bns[0].momentum is not None else 0.1
    sinks = [grad_sink(p) for p in params]
    nbt = [_nbt(bn) for bn in bns] if training else None   # bumped by the statistics kernels themselves
    cfg = (nsample, bool(group_all), bool(training), eps, momentum, sinks if any(s is not None for s in sinks) else None, nbt, pre)
    new_xyz, out, nbr = _SetAbstraction.apply(xyz, points, centre_idx, neighbour_idx, cfg, running, *params)
    if return_neighbours:
        return new_xyz, out, (None if group_all else nbr)
    return new_xyz, out


# ------------------------------------------------------------------------------------------------
# fully connected block
# ------------------------------------------------------------------------------------------------
class _FcBlock(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, nw, nb, rm, rv, mask, cfg):
        norm, relu, training, eps, momentum, drop_scale, sinks, nbt, draw = cfg
        x, w, b = _f32(x, "x"), _f32(w, "weight"), _f32(b, "bias")
        M, K = x.shape
        N = w.shape[0]
        d = L.FcDesc()
        d.M, d.K, d.N, d.norm, d.relu, d.training = M, K, N, norm, int(relu), int(training)
        d.eps, d.momentum, d.drop_scale = float(eps), float(momentum), float(drop_scale)
        lib = L.lib()
        sb = lib.pnpp_fc_saved_bytes(C.byref(d))
        if sb == 0:
            L.check(L.PNPP_ERR_ARG)
        saved = torch.empty(sb, dtype=torch.uint8, device=x.device)
        scratch = _scratch(lib.pnpp_fc_scratch_bytes(C.byref(d)), x.device)
        y = torch.empty(M, N, device=x.device, dtype=torch.float32)
        if nw is not None:
            nw, nb = _f32(nw, "norm weight"), _f32(nb, "norm bias")
        if mask is not None:
            mask = mask.contiguous()
        a = L.FcFwdArgs()
        a.x, a.w, a.b, a.nw, a.nb = x.data_ptr(), w.data_ptr(), b.data_ptr(), _p(nw), _p(nb)
        a.rm, a.rv, a.nbt, a.mask = _p(rm), _p(rv), _p(nbt), _p(mask)
        if draw is not None:   # the kernel draws the keep-mask itself (no RNG launch, graph-replayable) and hands it back
            p_drop, seed, counter = draw
            mask = torch.empty(M, N, device=x.device, dtype=torch.uint8)
            a.mask_out, a.drop_p, a.rng_seed, a.rng_counter = mask.data_ptr(), float(p_drop), int(seed) & (2**64 - 1), counter.data_ptr()
        a.y, a.saved, a.scratch = y.data_ptr(), saved.data_ptr(), scratch.data_ptr()
        L.check(lib.pnpp_fc_forward(C.byref(d), C.byref(a), _stream()))
        ctx.desc = d
        ctx.sinks = sinks
        ctx.has_norm, ctx.has_mask = nw is not None, mask is not None
        e = x.new_empty(0)
        ctx.save_for_backward(x, w, b, nw if nw is not None else e, nb if nb is not None else e,
                              mask if mask is not None else e, saved)
        return y

    @staticmethod
    def backward(ctx, dy):
        d = ctx.desc
        x, w, b, nw, nb, mask, saved = ctx.saved_tensors
        nw, nb = (nw, nb) if ctx.has_norm else (None, None)
        mask = mask if ctx.has_mask else None
        dy = _f32(dy, "dy")
        lib = L.lib()
        scratch = _scratch(lib.pnpp_fc_scratch_bytes(C.byref(d)), x.device)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        sk = ctx.sinks or (None, None, None, None)
        dw = sk[0] if sk[0] is not None else torch.empty_like(w)
        db = sk[1] if sk[1] is not None else torch.empty_like(b)
        dnw = (sk[2] if sk[2] is not None else torch.empty_like(nw)) if nw is not None else None
        dnb = (sk[3] if sk[3] is not None else torch.empty_like(nb)) if nb is not None else None
        a = L.FcBwdArgs()
        a.x, a.w, a.b, a.nw, a.nb, a.mask = x.data_ptr(), w.data_ptr(), b.data_ptr(), _p(nw), _p(nb), _p(mask)
        a.dy, a.saved, a.scratch = dy.data_ptr(), saved.data_ptr(), scratch.data_ptr()
        a.dx, a.dw, a.db, a.dnw, a.dnb = _p(dx), dw.data_ptr(), db.data_ptr(), _p(dnw), _p(dnb)
        L.check(lib.pnpp_fc_backward(C.byref(d), C.byref(a), _stream()))
        keep = lambda t, i: None if sk[i] is not None else t
        return dx, keep(dw, 0), keep(db, 1), keep(dnw, 2), keep(dnb, 3), None, None, None, None


def fc_block(x, linear, norm=None, relu=False, dropout=None, training=True, mask=None):
    """Linear -> optional BatchNorm1d/LayerNorm -> optional ReLU -> optional dropout.

    `dropout` is an nn.Dropout (its p is used, a fresh keep-mask is drawn on the device in training) or
    None; `mask` injects an explicit (M,N) keep-mask instead (tests / parity runs).
    """
    import torch.nn as nn
    nw = nb = rm = rv = None
    kind, eps, momentum = L.NORM_NONE, 1e-5, 0.1
    if isinstance(norm, nn.BatchNorm1d):
        kind, eps = L.NORM_BATCH, norm.eps
        momentum = norm.momentum if norm.momentum is not None else 0.1
        nw, nb, rm, rv = norm.weight, norm.bias, norm.running_mean, norm.running_var
    elif isinstance(norm, nn.LayerNorm):
        kind, eps = L.NORM_LAYER, norm.eps
        nw, nb = norm.weight, norm.bias
    elif norm is not None:
        raise TypeError(f"unsupported norm module {type(norm).__name__}")
    drop_scale = 1.0
    draw = None
    if training and mask is None and dropout is not None and dropout.p > 0:
        from . import dist as _pdist
        if x.is_cuda and ((kind == L.NORM_BATCH and x.shape[0] <= 32 and not _pdist.sync_batchnorm_enabled()) or
                          (kind == L.NORM_LAYER and dropout.p < 1.0)):
            # the BatchNorm epilogue / the LayerNorm pass draws it in the kernel (SyncBN takes the unfused route: the mask comes from torch)
            draw = (dropout.p, torch.initial_seed(), _dropout_counter(x.device))
            drop_scale = 1.0 / (1.0 - dropout.p)
        else:
            mask = torch.empty(x.shape[0], linear.weight.shape[0], device=x.device, dtype=torch.uint8).bernoulli_(1.0 - dropout.p)
    if mask is not None:
        if not training:
            mask = None
        else:
            p = dropout.p if dropout is not None else 0.5
            drop_scale = 1.0 / (1.0 - p)
            mask = mask.to(device=x.device, dtype=torch.uint8)
    sinks = tuple(grad_sink(p) for p in (linear.weight, linear.bias, nw, nb))
    nbt = _nbt(norm) if (training and kind == L.NORM_BATCH) else None   # bumped by the statistics kernel itself
    cfg = (kind, relu, training, eps, momentum, drop_scale, sinks if any(s is not None for s in sinks) else None, nbt, draw)
    return _FcBlock.apply(x, linear.weight, linear.bias, nw, nb, rm, rv, mask, cfg)


# ------------------------------------------------------------------------------------------------
# heads and losses
# ------------------------------------------------------------------------------------------------
class _VmHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, o):
        o = _f32(o, "o")
        B = o.shape[0]
        mu = torch.empty(B, device=o.device, dtype=torch.float32)
        kappa = torch.empty_like(mu)
        L.check(L.lib().pnpp_vm_head_kl(o.data_ptr(), None, None, B, mu.data_ptr(), kappa.data_ptr(), None, None, _stream()))
        ctx.save_for_backward(o)
        return mu, kappa

    @staticmethod
    def backward(ctx, dmu, dkappa):
        (o,) = ctx.saved_tensors
        B = o.shape[0]
        dmu = _f32(dmu, "dmu") if dmu is not None else torch.zeros(B, device=o.device)
        dkappa = _f32(dkappa, "dkappa") if dkappa is not None else torch.zeros(B, device=o.device)
        d_o = torch.empty_like(o)
        L.check(L.lib().pnpp_vm_head_bwd(o.data_ptr(), dmu.data_ptr(), dkappa.data_ptr(), B, d_o.data_ptr(), _stream()))
        return d_o


def vm_head(o: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """mu = tanh(o[:,0])*pi, kappa = softplus(o[:,1])  (pointnet_pp_vonMises.py:36-37)."""
    return _VmHead.apply(o)


class _KlSingle(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu_p, kappa_p, mu_q, kappa_q):
        mu_p, kappa_p = _f32(mu_p, "mu_p"), _f32(kappa_p, "kappa_p")
        mu_q, kappa_q = _f32(mu_q, "mu_q"), _f32(kappa_q, "kappa_q")
        n = mu_p.numel()
        kl, dmu, dk = torch.empty_like(mu_p), torch.empty_like(mu_p), torch.empty_like(mu_p)
        L.check(L.lib().pnpp_vm_kl_single(mu_p.data_ptr(), kappa_p.data_ptr(), mu_q.data_ptr(), kappa_q.data_ptr(), n,
                                          kl.data_ptr(), dmu.data_ptr(), dk.data_ptr(), _stream()))
        ctx.save_for_backward(dmu, dk)
        return kl

    @staticmethod
    def backward(ctx, g):
        dmu, dk = ctx.saved_tensors
        return g * dmu, g * dk, None, None


def kl_von_mises_single(mu_p, kappa_p, mu_q, kappa_q):
    """train_single_peak_vonMises_KL.py:23-28 (value and analytic gradient from one launch)."""
    return _KlSingle.apply(mu_p, kappa_p, mu_q, kappa_q)


def vm_head_kl_fused(o, mu_gt, kappa_gt):
    """fc3 output -> (mu, kappa, loss_vec, d loss_vec/d o) in ONE launch (no autograd graph)."""
    o, mu_gt, kappa_gt = _f32(o, "o"), _f32(mu_gt, "mu_gt"), _f32(kappa_gt, "kappa_gt")
    B = o.shape[0]
    mu = torch.empty(B, device=o.device, dtype=torch.float32)
    kappa, lv, d_o = torch.empty_like(mu), torch.empty_like(mu), torch.empty_like(o)
    L.check(L.lib().pnpp_vm_head_kl(o.data_ptr(), mu_gt.data_ptr(), kappa_gt.data_ptr(), B, mu.data_ptr(), kappa.data_ptr(),
                                    lv.data_ptr(), d_o.data_ptr(), _stream()))
    return mu, kappa, lv, d_o


class _VmHeadKlLoss(torch.autograd.Function):
    """Raw fc3 output -> KL loss (per sample or batch mean) with the head, the loss and their gradient in one launch."""

    @staticmethod
    def forward(ctx, o, mu_gt, kappa_gt, mean):
        o, mu_gt, kappa_gt = _f32(o, "o"), _f32(mu_gt, "mu_gt"), _f32(kappa_gt, "kappa_gt")
        B = o.shape[0]
        if o.dim() != 2 or o.shape[1] != 2 or mu_gt.numel() != B or kappa_gt.numel() != B:
            raise ValueError(f"vm_head_kl_loss: o must be (B,2) and the targets (B,), got {tuple(o.shape)}, "
                             f"{tuple(mu_gt.shape)}, {tuple(kappa_gt.shape)}")
        d_o = torch.empty_like(o)
        ctx.mean = bool(mean)
        if mean:
            loss = torch.empty((), device=o.device, dtype=torch.float32)
            L.check(L.lib().pnpp_vm_head_kl_mean(o.data_ptr(), mu_gt.data_ptr(), kappa_gt.data_ptr(), B, None, None, None,
                                                 loss.data_ptr(), d_o.data_ptr(), _stream()))
        else:
            loss = torch.empty(B, device=o.device, dtype=torch.float32)
            mu, kappa = torch.empty_like(loss), torch.empty_like(loss)
            L.check(L.lib().pnpp_vm_head_kl(o.data_ptr(), mu_gt.data_ptr(), kappa_gt.data_ptr(), B, mu.data_ptr(),
                                            kappa.data_ptr(), loss.data_ptr(), d_o.data_ptr(), _stream()))
        ctx.save_for_backward(d_o)
        return loss

    @staticmethod
    def backward(ctx, g):
        (d_o,) = ctx.saved_tensors
        return (g * d_o if ctx.mean else g[:, None] * d_o), None, None, None


def vm_head_kl_loss(o, mu_gt, kappa_gt, reduction: str = "mean"):
    """KL(vM(head(o)) || vM(mu_gt, kappa_gt)) from the raw fc3 output `o` (B,2): pointnet_pp_vonMises.py:36-37 +
    train_single_peak_vonMises_KL.py:23-28 (+ the `.mean()` of line 83 for reduction="mean") in one launch."""
    if reduction not in ("mean", "none"):
        raise ValueError(f"reduction must be 'mean' or 'none', got {reduction!r}")
    return _VmHeadKlLoss.apply(o, mu_gt, kappa_gt, reduction == "mean")


def vm_head_kl_loss_backward(o, mu_gt, kappa_gt) -> torch.Tensor:
    """The training step's `loss = kl(...).mean(); loss.backward()` in one go: the fused launch already produces
    d loss / d o, so the backward pass is seeded with it directly (autograd would otherwise launch a fill for the
    scalar's unit gradient and a multiply by it).  Returns the detached mean loss; gradients of everything upstream of
    `o` are accumulated exactly as by loss.backward()."""
    o32, mu_gt, kappa_gt = _f32(o, "o"), _f32(mu_gt, "mu_gt"), _f32(kappa_gt, "kappa_gt")
    B = o32.shape[0]
    if o32.dim() != 2 or o32.shape[1] != 2 or mu_gt.numel() != B or kappa_gt.numel() != B:
        raise ValueError("vm_head_kl_loss_backward: o must be (B,2) and the targets (B,)")
    loss = torch.empty((), device=o32.device, dtype=torch.float32)
    d_o = torch.empty_like(o32)
    L.check(L.lib().pnpp_vm_head_kl_mean(o32.data_ptr(), mu_gt.data_ptr(), kappa_gt.data_ptr(), B, None, None, None,
                                         loss.data_ptr(), d_o.data_ptr(), _stream()))
    if o.requires_grad:
        torch.autograd.backward([o], [d_o])
    return loss


def vm_fc_head_kl_loss_backward(x, linear, mu_gt, kappa_gt, next_centres=None) -> torch.Tensor:
    """`o = linear(x)` (the model's fc3, two outputs), head, single-peak KL, `.mean()` and `loss.backward()` in ONE launch
    (pointnet_pp_vonMises.py:35-37 + train_single_peak_vonMises_KL.py:82-84): the gradients of `linear` land in its
    .grad (or the optimiser's flat buffer), the gradient of x seeds the rest of the backward pass.  Returns the
    detached mean loss.  Equals vm_head_kl_loss_backward(fc_block(x, linear), ...) to float32 rounding."""
    x32, mu_gt, kappa_gt = _f32(x, "x"), _f32(mu_gt, "mu_gt"), _f32(kappa_gt, "kappa_gt")
    w, b = _f32(linear.weight, "weight"), _f32(linear.bias, "bias")
    B, K = x32.shape
    if w.shape != (2, K) or b.numel() != 2 or mu_gt.numel() != B or kappa_gt.numel() != B:
        raise ValueError("vm_fc_head_kl_loss_backward: linear must map K -> 2 and the targets must be (B,)")
    if 4 * (2 * B + 2 * K + 8 + (B * K if B * K <= 12288 else 0)) > 56 * 1024:
        # the one-launch form stages fc3's weights and the outputs in LDS (56 KB); wider heads take the three launches it fuses
        if next_centres is not None:
            seed, counter, offset, Bs, N1, c1, N2, c2 = next_centres
            L.check(L.lib().pnpp_sample_random_dev2(int(seed) & (2**64 - 1), counter.data_ptr(), int(offset), Bs, N1, c1.shape[1],
                                                    c1.data_ptr(), N2, c2.shape[1], c2.data_ptr(), _stream()))
        return vm_head_kl_loss_backward(fc_block(x, linear, training=True), mu_gt, kappa_gt)
    loss = torch.empty((), device=x32.device, dtype=torch.float32)
    sinks = [grad_sink(p) for p in (linear.weight, linear.bias)]
    dw = sinks[0] if sinks[0] is not None else torch.empty_like(w)
    db = sinks[1] if sinks[1] is not None else torch.empty_like(b)
    dx = torch.empty_like(x32) if x.requires_grad else None
    if next_centres is not None:   # sampling.CentreRing.job(): the next step's centre draw rides in this launch's idle CUs
        seed, counter, offset, Bs, N1, c1, N2, c2 = next_centres
        L.check(L.lib().pnpp_vm_fc_head_kl_step_sample(x32.data_ptr(), w.data_ptr(), b.data_ptr(), mu_gt.data_ptr(), kappa_gt.data_ptr(),
                                                       B, K, loss.data_ptr(), dw.data_ptr(), db.data_ptr(), _p(dx), int(seed) & (2**64 - 1),
                                                       counter.data_ptr(), int(offset), Bs, N1, c1.shape[1], c1.data_ptr(), N2, c2.shape[1],
                                                       c2.data_ptr(), _stream()))
    else:
        L.check(L.lib().pnpp_vm_fc_head_kl_step(x32.data_ptr(), w.data_ptr(), b.data_ptr(), mu_gt.data_ptr(), kappa_gt.data_ptr(), B, K,
                                                loss.data_ptr(), dw.data_ptr(), db.data_ptr(), _p(dx), _stream()))
    for p, g, sink in ((linear.weight, dw, sinks[0]), (linear.bias, db, sinks[1])):
        if sink is None and p.requires_grad:                     # plain autograd semantics: accumulate into .grad
            if p.grad is None:
                p.grad = g.view_as(p)
            else:
                p.grad.add_(g.view_as(p))                        # in place: .grad may be a view of a flat buffer
    if dx is not None:
        torch.autograd.backward([x], [dx])
    return loss


def mvm_heads_match_loss_backward(x, head_pi, head_mu, head_kappa, vm_gt, K_gt, temp, kappa_max, next_centres=None, outputs=False):
    """The three output heads of PointNetPPMvM on features `x` (B,K), the head activations, `match_loss(...).mean()` and
    `loss.backward()` in ONE launch (models/pointnet_pp_mvM.py:91-125 + train_multi_peaks_vonMises_KL.py:54-81, :229-234): the
    heads' gradients land in their .grad (or the optimiser's flat buffer), the gradient of x seeds the rest of the backward pass.
    Returns the detached mean loss (and mu, kappa, weight with outputs=True).  Equals
    match_loss(*mvm_head(fc_block(x, head_pi), fc_block(x, head_mu), fc_block(x, head_kappa), ...), vm_gt, K_gt).mean() + backward()
    to float32 rounding; max_K other than 4 / 8 or heads too wide for the launch's LDS take exactly that route."""
    x32, vm_gt = _f32(x, "x"), _f32(vm_gt, "vm_gt")
    K32 = _i32(K_gt, "K_gt")
    B, K = x32.shape
    maxK = head_pi.weight.shape[0]
    heads = (head_pi, head_mu, head_kappa)
    fits = maxK in (4, 8) and K % 4 == 0 and 4 * (4 * maxK * (B + K) + B * K + 3 * B * maxK * (1 + maxK)) <= 96 * 1024
    if (head_mu.weight.shape[0] != 2 * maxK or head_kappa.weight.shape[0] != maxK or any(h.weight.shape[1] != K for h in heads)
            or tuple(vm_gt.shape) != (B, maxK, 3)):
        raise ValueError("mvm_heads_match_loss_backward: heads must map K -> max_K / 2 max_K / max_K and vm_gt must be (B, max_K, 3)")
    if not fits:
        if next_centres is not None:
            seed, counter, offset, Bs, N1, c1, N2, c2 = next_centres
            L.check(L.lib().pnpp_sample_random_dev2(int(seed) & (2**64 - 1), counter.data_ptr(), int(offset), Bs, N1, c1.shape[1],
                                                    c1.data_ptr(), N2, c2.shape[1], c2.data_ptr(), _stream()))
        mu, kappa, weight = mvm_head(*[fc_block(x, h, training=True) for h in heads], temp, kappa_max)
        loss = match_loss(mu, kappa, weight, vm_gt, K_gt).mean()
        loss.backward()
        loss = loss.detach()
        return (loss, mu.detach(), kappa.detach(), weight.detach()) if outputs else loss
    ws = [_f32(h.weight, "weight") for h in heads]
    bs = [_f32(h.bias, "bias") for h in heads]
    loss = torch.empty((), device=x32.device, dtype=torch.float32)
    sinks = [grad_sink(p) for h in heads for p in (h.weight, h.bias)]
    grads = [sinks[i] if sinks[i] is not None else torch.empty_like(t) for i, t in enumerate(t for pair in zip(ws, bs) for t in pair)]
    dx = torch.empty_like(x32) if x.requires_grad else None
    outs = [torch.empty(B, maxK, device=x32.device, dtype=torch.float32) for _ in range(3)] if outputs else [None] * 3
    if next_centres is not None:   # sampling.CentreRing.job(): the next step's centre draw rides in this launch's idle CUs
        seed, counter, offset, Bs, N1, c1, N2, c2 = next_centres
        samp = (int(seed) & (2**64 - 1), counter.data_ptr(), int(offset), Bs, N1, c1.shape[1], c1.data_ptr(), N2, c2.shape[1], c2.data_ptr())
    else:
        samp = (0, None, 0, 0, 0, 0, None, 0, 0, None)
    L.check(L.lib().pnpp_mvm_fc_head_match_step(x32.data_ptr(), ws[0].data_ptr(), bs[0].data_ptr(), ws[1].data_ptr(), bs[1].data_ptr(),
                                                ws[2].data_ptr(), bs[2].data_ptr(), vm_gt.data_ptr(), K32.data_ptr(), B, K, maxK, float(temp),
                                                float(kappa_max), loss.data_ptr(), grads[0].data_ptr(), grads[1].data_ptr(),
                                                grads[2].data_ptr(), grads[3].data_ptr(), grads[4].data_ptr(), grads[5].data_ptr(), _p(dx),
                                                _p(outs[0]), _p(outs[1]), _p(outs[2]), *samp, _stream()))
    params = [p for h in heads for p in (h.weight, h.bias)]
    for p, g, sink in zip(params, grads, sinks):
        if sink is None and p.requires_grad:                         # plain autograd semantics: accumulate into .grad
            if p.grad is None:
                p.grad = g.view_as(p)
            else:
                p.grad.add_(g.view_as(p))
    if dx is not None:
        torch.autograd.backward([x], [dx])
    return (loss, *outs) if outputs else loss


class _MatchLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, kappa, w, vm_gt, K_gt):
        mu, kappa, w, vm_gt = _f32(mu, "mu"), _f32(kappa, "kappa"), _f32(w, "w"), _f32(vm_gt, "vm_gt")
        K32 = _i32(K_gt, "K_gt")
        B, maxK = mu.shape
        lv = torch.empty(B, device=mu.device, dtype=torch.float32)
        dmu, dk, dw = torch.empty_like(mu), torch.empty_like(mu), torch.empty_like(mu)
        assign = torch.empty(B, maxK, device=mu.device, dtype=torch.int32)
        L.check(L.lib().pnpp_vm_match_loss(mu.data_ptr(), kappa.data_ptr(), w.data_ptr(), vm_gt.data_ptr(), K32.data_ptr(), B,
                                           maxK, lv.data_ptr(), dmu.data_ptr(), dk.data_ptr(), dw.data_ptr(),
                                           assign.data_ptr(), _stream()))
        ctx.save_for_backward(dmu, dk, dw)
        ctx.assign = assign
        return lv

    @staticmethod
    def backward(ctx, g):
        dmu, dk, dw = ctx.saved_tensors
        g = g[:, None]
        return g * dmu, g * dk, g * dw, None, None


def match_loss(mu, kappa, w, vm_gt, K_gt):
    """train_multi_peaks_vonMises_KL.py:54-81, whole batch in one launch, no host round trip."""
    return _MatchLoss.apply(mu, kappa, w, vm_gt, K_gt)


class _MvmHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pi_raw, mu_raw, kappa_raw, temp, kappa_max):
        pi_raw, mu_raw, kappa_raw = _f32(pi_raw, "pi_raw"), _f32(mu_raw, "mu_raw"), _f32(kappa_raw, "kappa_raw")
        B, K = pi_raw.shape
        mu, kappa, weight = torch.empty_like(pi_raw), torch.empty_like(pi_raw), torch.empty_like(pi_raw)
        L.check(L.lib().pnpp_mvm_head(pi_raw.data_ptr(), mu_raw.data_ptr(), kappa_raw.data_ptr(), B, K, float(temp),
                                      float(kappa_max), mu.data_ptr(), kappa.data_ptr(), weight.data_ptr(), _stream()))
        ctx.save_for_backward(pi_raw, mu_raw, kappa_raw, weight)
        ctx.cfg = (float(temp), float(kappa_max))
        return mu, kappa, weight

    @staticmethod
    def backward(ctx, dmu, dkappa, dweight):
        pi_raw, mu_raw, kappa_raw, weight = ctx.saved_tensors
        B, K = pi_raw.shape
        z = lambda t: _f32(t, "grad") if t is not None else torch.zeros_like(pi_raw)
        dmu, dkappa, dweight = z(dmu), z(dkappa), z(dweight)
        dpi, dmr, dkr = torch.empty_like(pi_raw), torch.empty_like(mu_raw), torch.empty_like(kappa_raw)
        L.check(L.lib().pnpp_mvm_head_bwd(pi_raw.data_ptr(), mu_raw.data_ptr(), kappa_raw.data_ptr(), weight.data_ptr(),
                                          dmu.data_ptr(), dkappa.data_ptr(), dweight.data_ptr(), B, K, ctx.cfg[0], ctx.cfg[1],
                                          dpi.data_ptr(), dmr.data_ptr(), dkr.data_ptr(), _stream()))
        return dpi, dmr, dkr, None, None


def mvm_head(pi_raw, mu_raw, kappa_raw, temp: float, kappa_max: Optional[float]):
    return _MvmHead.apply(pi_raw, mu_raw, kappa_raw, temp, float("inf") if kappa_max is None else kappa_max)


class _SoftCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, p):
        logits, p = _f32(logits, "logits"), _f32(p, "p_target")
        B, Cc = logits.shape
        lv, dl = torch.empty(B, device=logits.device, dtype=torch.float32), torch.empty_like(logits)
        L.check(L.lib().pnpp_soft_ce(logits.data_ptr(), p.data_ptr(), B, Cc, lv.data_ptr(), dl.data_ptr(), _stream()))
        ctx.save_for_backward(dl)
        return lv

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return g[:, None] * dl, None


def soft_ce(logits, p_target):
    """train_8dir_KL.py:60-68."""
    return _SoftCE.apply(logits, p_target)


# ------------------------------------------------------------------------------------------------
# direction-vector heads and losses of the other set-abstraction models
# ------------------------------------------------------------------------------------------------
class _L2Normalize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps):
        x = _f32(x, "x")
        if x.dim() != 2:
            raise ValueError(f"l2_normalize: expected (M,C), got {tuple(x.shape)}")
        M, Cc = x.shape
        y = torch.empty_like(x)
        L.check(L.lib().pnpp_l2_normalize(x.data_ptr(), M, Cc, float(eps), y.data_ptr(), _stream()))
        ctx.save_for_backward(x)
        ctx.eps = float(eps)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = _f32(dy, "dy")
        dx = torch.empty_like(x)
        L.check(L.lib().pnpp_l2_normalize_bwd(x.data_ptr(), dy.data_ptr(), x.shape[0], x.shape[1], ctx.eps, dx.data_ptr(),
                                              _stream()))
        return dx, None


def l2_normalize(x: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    """F.normalize(x, p=2, dim=1, eps) for (M,C) inputs (pointnet_pp_Fwd.py:98, Pointnet_pp_xyz.py:84-85)."""
    return _L2Normalize.apply(x, eps)


class _Mse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, t):
        p, t = _f32(p, "input"), _f32(t, "target")
        if p.shape != t.shape:
            raise ValueError(f"mse_loss: input {tuple(p.shape)} and target {tuple(t.shape)} differ")
        loss = torch.empty((), device=p.device, dtype=torch.float32)
        dp = torch.empty_like(p)
        L.check(L.lib().pnpp_mse(p.data_ptr(), t.data_ptr(), p.numel(), loss.data_ptr(), dp.data_ptr(), _stream()))
        ctx.save_for_backward(dp)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dp,) = ctx.saved_tensors
        return g * dp, None


def mse_loss(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """nn.MSELoss()(pred, target): mean over all elements, value and gradient from one launch."""
    return _Mse.apply(pred, target)


class _MseRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, t):
        p, t = _f32(p, "input"), _f32(t, "target")
        if p.shape != t.shape or p.dim() != 2:
            raise ValueError(f"mse_rows: expected two (B,C) tensors, got {tuple(p.shape)} and {tuple(t.shape)}")
        lv, dp = torch.empty(p.shape[0], device=p.device, dtype=torch.float32), torch.empty_like(p)
        L.check(L.lib().pnpp_mse_rows(p.data_ptr(), t.data_ptr(), p.shape[0], p.shape[1], lv.data_ptr(), dp.data_ptr(), _stream()))
        ctx.save_for_backward(dp)
        return lv

    @staticmethod
    def backward(ctx, g):
        (dp,) = ctx.saved_tensors
        return g[:, None] * dp, None


def mse_rows(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """Per-sample mean squared error (B,C) -> (B,) (simple_pointnet_train.py:174); its mean is nn.MSELoss()(pred, target)."""
    return _MseRows.apply(pred, target)


class _Orth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _f32(a, "a"), _f32(b, "b")
        if a.shape != b.shape or a.dim() != 2:
            raise ValueError(f"orth_loss: expected two (B,C) tensors, got {tuple(a.shape)} and {tuple(b.shape)}")
        loss = torch.empty((), device=a.device, dtype=torch.float32)
        da, db = torch.empty_like(a), torch.empty_like(b)
        L.check(L.lib().pnpp_orth_loss(a.data_ptr(), b.data_ptr(), a.shape[0], a.shape[1], loss.data_ptr(), da.data_ptr(),
                                       db.data_ptr(), _stream()))
        ctx.save_for_backward(da, db)
        return loss

    @staticmethod
    def backward(ctx, g):
        da, db = ctx.saved_tensors
        return g * da, g * db


def orth_loss(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """(a * b).sum(1).pow(2).mean()  (train.py:184-185)."""
    return _Orth.apply(a, b)


class _ProjProbs(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vec, dirs):
        vec, dirs = _f32(vec, "vec"), _f32(dirs, "dirs")
        if vec.dim() != 2 or vec.shape[1] != 3 or dirs.dim() != 2 or dirs.shape[1] != 3:
            raise ValueError(f"proj_probs: expected vec (B,3) and dirs (D,3), got {tuple(vec.shape)}, {tuple(dirs.shape)}")
        B, D = vec.shape[0], dirs.shape[0]
        probs = torch.empty(B, D, device=vec.device, dtype=torch.float32)
        L.check(L.lib().pnpp_proj_probs(vec.data_ptr(), dirs.data_ptr(), B, D, probs.data_ptr(), _stream()))
        ctx.save_for_backward(vec, dirs)
        return probs

    @staticmethod
    def backward(ctx, dprobs):
        vec, dirs = ctx.saved_tensors
        dprobs = _f32(dprobs, "dprobs")
        dvec = torch.empty_like(vec)
        L.check(L.lib().pnpp_proj_probs_bwd(vec.data_ptr(), dirs.data_ptr(), dprobs.data_ptr(), vec.shape[0], dirs.shape[0],
                                            dvec.data_ptr(), _stream()))
        return dvec, None


def proj_probs(vec: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """train_multi_8dir.py:41-44: unit vector -> non-negative cosine to each of the D directions -> normalised to sum 1."""
    return _ProjProbs.apply(vec, dirs)
