"""oracle/restatement.py -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's PointNet++ set-abstraction + von-Mises-KL hot path.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the shipped path (3d-pointcloud-orientation-estimation_amd/) never does and fails
loudly when its HIP library is missing.

What is restated (paths relative to /root/reference):
  index primitives ............ models/base.py:4-35, PointNet++Demo.py:8-70  (C, oracle/index_ops.c)
  PointNetSetAbstraction ....... models/pointnet_pp_8dir.py:6-43
  PointNetPPVonMises ........... models/pointnet_pp_vonMises.py:8-38
  PointNetPPMvM ................ models/pointnet_pp_mvM.py:30-127
  PointNetPP8Dir ............... models/pointnet_pp_8dir.py:58-85
  single-peak kl_von_mises ..... train_single_peak_vonMises_KL.py:23-28
  multi-peak kl + match_loss ... train_multi_peaks_vonMises_KL.py:38-81
  soft-label CE ................ train_8dir_KL.py:60-68

The restatement is functional (a state_dict in, tensors out) and dtype-generic: run it in
float32 to mirror the reference's CPU path, or in float64 to obtain the well-conditioned
value that SURVEY.md section 7a shows is the only 1e-5-reproducible yardstick.  Neighbour
sets always come from the float32 distance recipe, whatever the compute dtype.

Pinned against the imported reference by tests/golden/*.npz (oracle/make_golden.py).
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_index.so")
_lib = None


def build_c_oracle(force: bool = False) -> str:
    """Compile oracle/index_ops.c (gcc) if the shared object is missing."""
    if force or not os.path.exists(_SO):
        subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    return _SO


def _c():
    global _lib
    if _lib is None:
        build_c_oracle()
        lib = ctypes.CDLL(_SO)
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int32)
        ci = ctypes.c_int
        lib.oracle_square_distance.argtypes = [fp, fp, ci, ci, ci, fp]
        lib.oracle_square_distance.restype = None
        lib.oracle_knn.argtypes = [fp, fp, ci, ci, ci, ci, ip]
        lib.oracle_knn.restype = ci
        lib.oracle_fps.argtypes = [fp, ci, ci, ci, ip, ip]
        lib.oracle_fps.restype = None
        lib.oracle_ball_query.argtypes = [fp, fp, ci, ci, ci, ctypes.c_float, ci, ip]
        lib.oracle_ball_query.restype = None
        _lib = lib
    return _lib


def _f32(t) -> np.ndarray:
    if isinstance(t, torch.Tensor):
        t = t.detach().cpu().numpy()
    return np.ascontiguousarray(t, dtype=np.float32)


def _fp(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _ip(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


# --------------------------------------------------------------------------------------
# index primitives (bit-exact, C)
# --------------------------------------------------------------------------------------
def square_distance(src, dst) -> torch.Tensor:
    """models/base.py:20-27, float32, bit-exact to the ATen CPU evaluation order."""
    a, b = _f32(src), _f32(dst)
    B, S, _ = a.shape
    N = b.shape[1]
    out = np.empty((B, S, N), np.float32)
    _c().oracle_square_distance(_fp(a), _fp(b), B, S, N, _fp(out))
    return torch.from_numpy(out)


def knn_indices(new_xyz, xyz, k: int) -> torch.Tensor:
    """models/base.py:29-35: the k nearest by the float32 recipe; ascending (distance, index)."""
    a, b = _f32(new_xyz), _f32(xyz)
    B, S, _ = a.shape
    N = b.shape[1]
    out = np.empty((B, S, k), np.int32)
    if _c().oracle_knn(_fp(a), _fp(b), B, S, N, k, _ip(out)) != 0:
        raise RuntimeError("selected index k out of range")  # what topk raises in the reference
    return torch.from_numpy(out.astype(np.int64))


def farthest_point_sample(xyz, npoint: int, start) -> torch.Tensor:
    """PointNet++Demo.py:8-29 with the `torch.randint` start indices injected."""
    a = _f32(xyz)
    B, N, _ = a.shape
    st = np.ascontiguousarray(np.asarray(start), dtype=np.int32)
    out = np.empty((B, npoint), np.int32)
    _c().oracle_fps(_fp(a), B, N, npoint, _ip(st), _ip(out))
    return torch.from_numpy(out.astype(np.int64))


def ball_query(radius: float, nsample: int, xyz, new_xyz) -> torch.Tensor:
    """PointNet++Demo.py:49-70 (argument order as in the demo)."""
    b, a = _f32(xyz), _f32(new_xyz)
    B, N, _ = b.shape
    S = a.shape[1]
    out = np.empty((B, S, nsample), np.int32)
    _c().oracle_ball_query(_fp(a), _fp(b), B, S, N, float(radius), nsample, _ip(out))
    return torch.from_numpy(out.astype(np.int64))


def index_points(points: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """models/base.py:4-18: points (B,N,C), idx (B,S) or (B,S,K) -> rows of `points`."""
    B, _, C = points.shape
    flat = idx.reshape(B, -1, 1).expand(-1, -1, C)
    return torch.gather(points, 1, flat).reshape(*idx.shape, C)


# --------------------------------------------------------------------------------------
# set abstraction (pointnet_pp_8dir.py:6-43), channels-last functional form
# --------------------------------------------------------------------------------------
class BNState:
    """Collects the running-stat updates a train-mode forward performs (momentum 0.1)."""

    def __init__(self):
        self.updates: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}


class _SumOverRanks(torch.autograd.Function):
    """y = sum over ranks of x (every rank gets y); backward: the incoming gradients summed over the ranks -- what
    torch.nn.SyncBatchNorm does with its (sum, sum of squares, count) forward and (sum dy, sum dy xhat) backward."""

    @staticmethod
    def forward(ctx, x, fn):
        ctx.fn = fn
        return fn(x.detach().clone())

    @staticmethod
    def backward(ctx, g):
        return ctx.fn(g.detach().clone()), None


_STATS_SYNC = None   # callable(tensor) -> tensor summed over the ranks in place, or None (per-rank statistics, DDP's default)


class stats_sync:
    """with stats_sync(fn): every train-mode BatchNorm below pools its statistics over the ranks (SyncBN): a batch
    split over ranks then normalises like the reference's single process on the concatenated batch."""

    def __init__(self, fn):
        self.fn = fn

    def __enter__(self):
        global _STATS_SYNC
        self.prev, _STATS_SYNC = _STATS_SYNC, self.fn

    def __exit__(self, *exc):
        global _STATS_SYNC
        _STATS_SYNC = self.prev


def _bn_train(z: torch.Tensor, gamma, beta, dims, eps: float):
    if _STATS_SYNC is None:
        mean = z.mean(dim=dims)
        var = z.var(dim=dims, unbiased=False)
    else:   # biased variance from pooled (sum, sum of squares, rows): E[z^2] - E[z]^2 over the rows of every rank
        rows = z.numel() // z.shape[-1]
        packed = torch.cat([z.sum(dim=dims), (z * z).sum(dim=dims), torch.full((1,), float(rows), dtype=z.dtype)])
        packed = _SumOverRanks.apply(packed, _STATS_SYNC)
        C = z.shape[-1]
        mean = packed[:C] / packed[2 * C]
        var = (packed[C:2 * C] / packed[2 * C] - mean * mean).clamp_min(0.0)
    zh = (z - mean) / torch.sqrt(var + eps)
    return zh * gamma + beta, mean, var


def _pooled_rows(rows: int) -> int:
    """Rows behind a BatchNorm's statistics: this rank's, or every rank's under stats_sync."""
    if _STATS_SYNC is None:
        return rows
    return int(_STATS_SYNC(torch.tensor([float(rows)], dtype=torch.float64)).item())


def _bn_eval(z, gamma, beta, rm, rv, eps):
    return (z - rm) / torch.sqrt(rv + eps) * gamma + beta


def sa_forward(xyz32: torch.Tensor, points: Optional[torch.Tensor], P: Dict[str, torch.Tensor],
               prefix: str, centre_idx: Optional[torch.Tensor], nsample: Optional[int],
               group_all: bool, training: bool = True, bn_state: Optional[BNState] = None,
               eps: float = 1e-5, momentum: float = 0.1, neighbour_idx: Optional[torch.Tensor] = None,
               rel_in_compute_dtype: bool = False, argmax: Optional[torch.Tensor] = None, diag: Optional[dict] = None,
               relu_masks: Optional[Sequence[torch.Tensor]] = None):
    """One PointNetSetAbstraction.forward (pointnet_pp_8dir.py:21-43).

    xyz32 is the float32 cloud (B,N,3); compute dtype is that of the parameters in P.
    centre_idx (B,S) int64 replaces the `torch.randperm` draw of line 28.
    Returns new_xyz (B,S,3) float32, features (B,S,Cout) and the neighbour indices used.

    argmax (B,S,Cout) int64, optional: positions inside each neighbourhood (in the order of `neighbour_idx`) that the max
    over nsample takes -- the routing of another evaluation of the same network.  max() is not smooth: where two rows of
    a neighbourhood agree to float32 rounding, a float32 evaluation may legitimately route the pooled gradient through
    the other row than float64 does.  With the routing injected the result is a smooth function of rounding again;
    diag["route_gap"] collects max (true max - value at the injected position) / scale per call, which the caller bounds
    (the injected routing must select a maximum up to float32 rounding).

    relu_masks, optional: per layer a (B,S,K,C_l) {0,1} tensor -- the ReLU decisions of another evaluation.  ReLU's derivative is
    not smooth either: a pre-activation within float32 rounding of zero passes the gradient in one evaluation and blocks it in
    the other (an O(1) change of that element's gradient, ~1/sqrt(elements) of the layer's gradient norm).  With the decisions
    injected, layer l computes y * mask instead of relu(y) (they differ by the rounding-sized value itself, in forward and
    backward alike); diag["relu_flips"] collects, per call and layer, how many decisions differ from this evaluation's own and
    diag["relu_flip_margin"] the largest |y| / max|y| among them (which the caller bounds: an injected decision may only
    differ where the pre-activation is zero up to rounding)."""
    dt = P[f"{prefix}.convs.0.weight"].dtype
    B, N, _ = xyz32.shape
    if group_all:
        new_xyz32 = torch.zeros(B, 1, 3)
        g = xyz32.to(dt).unsqueeze(1)
        x = g if points is None else torch.cat([g, points.unsqueeze(1)], -1)
        idx = None
    else:
        new_xyz32 = index_points(xyz32, centre_idx)
        idx = neighbour_idx if neighbour_idx is not None else knn_indices(new_xyz32, xyz32, nsample)
        if rel_in_compute_dtype:   # what the reference does when its modules are run after .double()
            rel = index_points(xyz32.to(dt), idx) - new_xyz32.to(dt).unsqueeze(2)
        else:                      # fp32 subtraction as in :32 (the shipped kernels do this)
            rel = (index_points(xyz32, idx) - new_xyz32.unsqueeze(2)).to(dt)
        x = rel if points is None else torch.cat([rel, index_points(points, idx)], -1)
    li = 0
    while f"{prefix}.convs.{li}.weight" in P:
        W = P[f"{prefix}.convs.{li}.weight"]
        W = W.reshape(W.shape[0], -1)
        z = x @ W.t() + P[f"{prefix}.convs.{li}.bias"]
        g_, b_ = P[f"{prefix}.bns.{li}.weight"], P[f"{prefix}.bns.{li}.bias"]
        if training:
            y, mean, var = _bn_train(z, g_, b_, (0, 1, 2), eps)
            if bn_state is not None:
                m = _pooled_rows(z.numel() // z.shape[-1])
                rm, rv = P[f"{prefix}.bns.{li}.running_mean"], P[f"{prefix}.bns.{li}.running_var"]
                bn_state.updates[f"{prefix}.bns.{li}"] = (
                    ((1 - momentum) * rm + momentum * mean).detach(),
                    ((1 - momentum) * rv + momentum * var * (m / max(m - 1, 1))).detach())
        else:
            y = _bn_eval(z, g_, b_, P[f"{prefix}.bns.{li}.running_mean"], P[f"{prefix}.bns.{li}.running_var"], eps)
        if relu_masks is not None and relu_masks[li] is not None:
            m = relu_masks[li].to(y.device).reshape(y.shape) != 0
            if diag is not None:
                flip = m != (y.detach() > 0)
                diag.setdefault("relu_flips", []).append(int(flip.sum()))
                diag.setdefault("relu_flip_margin", []).append(
                    float((y.detach().abs() * flip).max() / y.detach().abs().max().clamp_min(1e-30)) if bool(flip.any()) else 0.0)
            x = y * m.to(dt)
        else:
            x = torch.relu(y)
        li += 1
    if argmax is None:
        return new_xyz32, x.max(dim=2).values, idx
    pooled = torch.gather(x, 2, argmax.to(torch.int64).unsqueeze(2)).squeeze(2)
    if diag is not None:
        top = x.detach().max(dim=2).values
        diag.setdefault("route_gap", []).append(float(((top - pooled.detach()) / top.abs().clamp_min(1.0)).max()))
        diag.setdefault("route_flips", []).append(int((pooled.detach() < top).sum()))   # injected positions that are not float64's maxima
    return new_xyz32, pooled, idx


def backbone_forward(xyz32, P, centres: Sequence[torch.Tensor], training=True, bn_state=None,
                     cfg=((128, 32), (32, 32)), rel_in_compute_dtype=False, routing=None, diag=None):
    """sa1 -> sa2 -> sa3(group_all), as in every pointnet_pp_* model (e.g. pointnet_pp_vonMises.py:28-31).
    routing: optional three dicts {"neighbours": (B,S,K) | None, "argmax": (B,S,C), "relu_masks": [(B,S,K,C_l)] (optional)}
    (see sa_forward)."""
    kw = dict(rel_in_compute_dtype=rel_in_compute_dtype, diag=diag)
    r = routing if routing is not None else [{"neighbours": None, "argmax": None}] * 3
    nb = lambda i: None if r[i]["neighbours"] is None else r[i]["neighbours"].to(torch.int64)
    rm = lambda i: r[i].get("relu_masks")
    l1_xyz, l1, _ = sa_forward(xyz32, None, P, "sa1", centres[0], cfg[0][1], False, training, bn_state,
                               neighbour_idx=nb(0), argmax=r[0]["argmax"], relu_masks=rm(0), **kw)
    l2_xyz, l2, _ = sa_forward(l1_xyz, l1, P, "sa2", centres[1], cfg[1][1], False, training, bn_state,
                               neighbour_idx=nb(1), argmax=r[1]["argmax"], relu_masks=rm(1), **kw)
    _, l3, _ = sa_forward(l2_xyz, l2, P, "sa3", None, None, True, training, bn_state, argmax=r[2]["argmax"], relu_masks=rm(2), **kw)
    return l3.reshape(l3.shape[0], -1)


def _bn1d(x, P, name, training, bn_state, eps=1e-5, momentum=0.1):
    g_, b_ = P[f"{name}.weight"], P[f"{name}.bias"]
    if training:
        y, mean, var = _bn_train(x, g_, b_, (0,), eps)
        if bn_state is not None:
            m = _pooled_rows(x.shape[0])
            bn_state.updates[name] = (
                ((1 - momentum) * P[f"{name}.running_mean"] + momentum * mean).detach(),
                ((1 - momentum) * P[f"{name}.running_var"] + momentum * var * (m / max(m - 1, 1))).detach())
        return y
    return _bn_eval(x, g_, b_, P[f"{name}.running_mean"], P[f"{name}.running_var"], eps)


def _lin(x, P, name):
    return x @ P[f"{name}.weight"].t() + P[f"{name}.bias"]


def bn_head_features(feat, P, drop_mask: Optional[torch.Tensor], training=True, bn_state=None, p_drop=0.5):
    """fc1/bn1/relu, fc2/bn2/relu, dropout (pointnet_pp_vonMises.py:32-34 and every other BN-head model).
    drop_mask (B,256) of {0,1} replaces nn.Dropout's draw; None = no dropout."""
    x = torch.relu(_bn1d(_lin(feat, P, "fc1"), P, "bn1", training, bn_state))
    x = torch.relu(_bn1d(_lin(x, P, "fc2"), P, "bn2", training, bn_state))
    if training and drop_mask is not None:
        x = x * drop_mask.to(x.dtype) / (1.0 - p_drop)
    return x


def bn_head_forward(feat, P, drop_mask: Optional[torch.Tensor], training=True, bn_state=None, p_drop=0.5):
    """... followed by fc3 (pointnet_pp_vonMises.py:35, pointnet_pp_8dir.py:85, pointnet_pp.py:68)."""
    return _lin(bn_head_features(feat, P, drop_mask, training, bn_state, p_drop), P, "fc3")


def vonmises_forward(xyz32, P, centres, drop_mask=None, training=True, bn_state=None, **bk):
    """PointNetPPVonMises.forward (pointnet_pp_vonMises.py:26-38) -> (mu, kappa)."""
    out = bn_head_forward(backbone_forward(xyz32, P, centres, training, bn_state, **bk), P, drop_mask, training, bn_state)
    return torch.tanh(out[:, 0]) * math.pi, F.softplus(out[:, 1])


def dir8_forward(xyz32, P, centres, drop_mask=None, training=True, bn_state=None, **bk):
    """PointNetPP8Dir.forward (pointnet_pp_8dir.py:76-85) -> logits (B,8)."""
    return bn_head_forward(backbone_forward(xyz32, P, centres, training, bn_state, **bk), P, drop_mask, training, bn_state)


# --------------------------------------------------------------------------------------
# The same step through the STOCK ATen ops the reference's modules call (cpu_baseline of bench.py only)
# --------------------------------------------------------------------------------------
# The functions above spell BatchNorm as elementwise tensor expressions and the 1x1 convolution as a matmul so that they
# run in any dtype and expose every intermediate; on a CPU that costs several passes over each activation.  The reference
# itself runs nn.Conv2d / nn.BatchNorm2d / F.relu / torch.max on a (B,C,npoint,nsample) tensor and dist.topk for the
# neighbour search, i.e. ATen's fused kernels.  These two functions are that: the representative CPU baseline.
def sa_forward_aten(xyz, points, P, prefix, centre_idx, nsample, group_all, momentum=0.1, eps=1e-5):
    """PointNetSetAbstraction.forward with the reference's own op sequence (models/pointnet_pp_8dir.py:21-43, models/base.py:4-35):
    matmul-form square_distance + topk(sorted=False), advanced-indexing gathers, permute to (B,C,S,K), F.conv2d (1x1) ->
    F.batch_norm(training=True) -> F.relu per layer, torch.max over nsample.  Running statistics are updated in place on clones
    (the reference's modules own theirs); float32 only; centre_idx replaces the randperm draw of line 28."""
    B, N, _ = xyz.shape
    if group_all:
        new_xyz = torch.zeros(B, 1, 3)
        g = xyz.unsqueeze(1)
        new_points = g if points is None else torch.cat([g, points.unsqueeze(1)], -1)
    else:
        new_xyz = index_points(xyz, centre_idx)
        dist = -2 * torch.matmul(new_xyz, xyz.transpose(2, 1))                     # models/base.py:24-26
        dist += torch.sum(new_xyz ** 2, dim=-1).unsqueeze(-1)
        dist += torch.sum(xyz ** 2, dim=-1).unsqueeze(1)
        _, idx = dist.topk(nsample, dim=-1, largest=False, sorted=False)           # models/base.py:33-34
        normed = index_points(xyz, idx) - new_xyz.unsqueeze(2)
        new_points = normed if points is None else torch.cat([normed, index_points(points, idx)], -1)
    x = new_points.permute(0, 3, 1, 2)
    li = 0
    while f"{prefix}.convs.{li}.weight" in P:
        W, b = P[f"{prefix}.convs.{li}.weight"], P[f"{prefix}.convs.{li}.bias"]
        x = F.conv2d(x, W.reshape(W.shape[0], -1, 1, 1), b)
        rm = P[f"{prefix}.bns.{li}.running_mean"].detach().clone()
        rv = P[f"{prefix}.bns.{li}.running_var"].detach().clone()
        x = F.relu(F.batch_norm(x, rm, rv, P[f"{prefix}.bns.{li}.weight"], P[f"{prefix}.bns.{li}.bias"], True, momentum, eps))
        li += 1
    x = torch.max(x, 3)[0]
    return new_xyz, x.permute(0, 2, 1)


def vonmises_forward_aten(xyz, P, centres, drop_mask=None, p_drop=0.5):
    """PointNetPPVonMises.forward in train mode with stock ops (models/pointnet_pp_vonMises.py:26-38): float32 -> (mu, kappa)."""
    l1_xyz, l1 = sa_forward_aten(xyz, None, P, "sa1", centres[0], 32, False)
    l2_xyz, l2 = sa_forward_aten(l1_xyz, l1, P, "sa2", centres[1], 32, False)
    _, l3 = sa_forward_aten(l2_xyz, l2, P, "sa3", None, None, True)
    x = l3.reshape(l3.shape[0], -1)
    for fc, bn in (("fc1", "bn1"), ("fc2", "bn2")):
        x = F.linear(x, P[f"{fc}.weight"], P[f"{fc}.bias"])
        x = F.relu(F.batch_norm(x, P[f"{bn}.running_mean"].detach().clone(), P[f"{bn}.running_var"].detach().clone(),
                                P[f"{bn}.weight"], P[f"{bn}.bias"], True, 0.1, 1e-5))
    if drop_mask is not None:
        x = x * drop_mask.to(x.dtype) / (1.0 - p_drop)
    out = F.linear(x, P["fc3.weight"], P["fc3.bias"])
    return torch.tanh(out[:, 0]) * math.pi, F.softplus(out[:, 1])


def l2_normalize(x, eps=1e-12):
    """F.normalize(x, p=2, dim=1, eps) (pointnet_pp_Fwd.py:98)."""
    return x / x.norm(dim=1, keepdim=True).clamp_min(eps)


def pp_forward(xyz32, P, centres, drop_mask=None, training=True, bn_state=None, **bk):
    """PointNetPP.forward (pointnet_pp.py:58-68) -> raw (B,3)."""
    return bn_head_forward(backbone_forward(xyz32, P, centres, training, bn_state, **bk), P, drop_mask, training, bn_state)


def fwd_forward(xyz32, P, centres, drop_mask=None, training=True, bn_state=None, **bk):
    """PointNetPPFwd.forward (pointnet_pp_Fwd.py:89-98) -> unit (B,3)."""
    return l2_normalize(pp_forward(xyz32, P, centres, drop_mask, training, bn_state, **bk))


def axes_forward(xyz32, P, centres, heads, drop_mask=None, training=True, bn_state=None, **bk):
    """PointNetPPXYZ.forward (heads=("head_x","head_y"), Pointnet_pp_xyz.py:66-90) and PointNetPPXYZ_Schedmit.forward
    (heads=("head_y","head_z"), Pointnet_pp_xyz_Schedmit.py:68-90) -> two unit (B,3) vectors."""
    feat = bn_head_features(backbone_forward(xyz32, P, centres, training, bn_state, **bk), P, drop_mask, training, bn_state)
    return tuple(l2_normalize(_lin(feat, P, h)) for h in heads)


def simple_pointnet_forward(xyz32, P, drop_mask=None, training=True, bn_state=None, p_drop=0.3, argmax=None, diag=None):
    """SimplePointNet.forward of the reference's simple_pointnet_train.py:103-113: Conv1d/BatchNorm1d/ReLU x 3 over every
    point (conv1..3, bn1..3), max over the cloud (:110), relu(bn4(fc1)), dropout(0.3), fc2.  The per-point part is the
    whole-cloud (group_all) form of sa_forward on raw coordinates, so the max-pool routing can be injected the same way.
    drop_mask (B,128) of {0,1} replaces the dropout draw; None = no dropout."""
    Q = {}
    for i in range(3):
        for k in ("weight", "bias"):
            Q[f"enc.convs.{i}.{k}"] = P[f"conv{i + 1}.{k}"]
        for k in ("weight", "bias", "running_mean", "running_var"):
            Q[f"enc.bns.{i}.{k}"] = P[f"bn{i + 1}.{k}"]
    st = BNState() if bn_state is not None else None
    _, feat, _ = sa_forward(xyz32, None, Q, "enc", None, None, True, training, st, argmax=argmax, diag=diag)
    if bn_state is not None:
        for i in range(3):
            if f"enc.bns.{i}" in st.updates:
                bn_state.updates[f"bn{i + 1}"] = st.updates[f"enc.bns.{i}"]
    x = torch.relu(_bn1d(_lin(feat.reshape(feat.shape[0], -1), P, "fc1"), P, "bn4", training, bn_state))
    if training and drop_mask is not None:
        x = x * drop_mask.to(x.dtype) / (1.0 - p_drop)
    return _lin(x, P, "fc2")


def mse_rows(pred, target):
    """Per-sample line of simple_pointnet_train.py:174; the mean of the rows is nn.MSELoss() (:153,182)."""
    return ((pred - target) ** 2).mean(dim=1)


def mse(pred, target):
    """nn.MSELoss() (train.py:168)."""
    return ((pred - target) ** 2).mean()


def axis_pair_loss(vy, vz, gy, gz, lam=0.1):
    """train.py:183-187."""
    return (mse(vy, gy) + mse(vz, gz)) / 2.0 + lam * (vy * vz).sum(dim=1).pow(2).mean()


def proj_probs(vec, dirs):
    """train_multi_8dir.py:41-44."""
    sims = (l2_normalize(vec) @ dirs.t()).clamp(min=0)
    return sims / sims.sum(dim=1, keepdim=True).clamp(min=1e-8)


def _layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def point_transformer_forward(xyz, P, num_heads=4, depth=6, return_layers=False):
    """models/point_transformer.py:15-20 in eval mode (all dropouts are identities): input_proj, `depth` post-norm
    nn.TransformerEncoderLayer(d_model=64, nhead=4, dim_feedforward=2048, relu) blocks, mean over the points, fc_out.
    Restated from torch.nn.functional.multi_head_attention_forward / TransformerEncoderLayer.forward (norm_first=False):
        q,k,v = split(x W_in^T + b_in); per head: softmax(q k^T / sqrt(d_head)) v; concat heads; out_proj
        x = LN1(x + attn);  x = LN2(x + W2 relu(W1 x + b1) + b2)"""
    x = xyz.to(P["input_proj.weight"].dtype) @ P["input_proj.weight"].t() + P["input_proj.bias"]
    B, N, E = x.shape
    dh = E // num_heads
    layers = []
    for l in range(depth):
        pre = f"transformer.layers.{l}."
        qkv = x @ P[pre + "self_attn.in_proj_weight"].t() + P[pre + "self_attn.in_proj_bias"]
        q, k, v = (t.reshape(B, N, num_heads, dh).transpose(1, 2) for t in qkv.split(E, dim=-1))   # (B,H,N,dh)
        att = torch.softmax((q * (1.0 / math.sqrt(dh))) @ k.transpose(-1, -2), dim=-1)
        o = (att @ v).transpose(1, 2).reshape(B, N, E)
        o = o @ P[pre + "self_attn.out_proj.weight"].t() + P[pre + "self_attn.out_proj.bias"]
        x = _layer_norm(x + o, P[pre + "norm1.weight"], P[pre + "norm1.bias"])
        h = torch.relu(x @ P[pre + "linear1.weight"].t() + P[pre + "linear1.bias"])
        f = h @ P[pre + "linear2.weight"].t() + P[pre + "linear2.bias"]
        x = _layer_norm(x + f, P[pre + "norm2.weight"], P[pre + "norm2.bias"])
        layers.append(x)
    out = x.mean(dim=1) @ P["fc_out.weight"].t() + P["fc_out.bias"]
    return (out, layers) if return_layers else out


def mvm_forward(xyz32, P, centres, drop_masks=(None, None), training=True, bn_state=None,
                max_K=4, kappa_max=80.0, p_drop=0.4, temp=0.7, **bk):
    """PointNetPPMvM.forward (pointnet_pp_mvM.py:75-127) -> (mu, kappa, weight), each (B,K).
    xyz32 is (B,N,3); the (B,3,N) input form of :15-27 is a host-side transpose only."""
    x = backbone_forward(xyz32, P, centres, training, bn_state, **bk)
    for fc, ln, mask in (("fc1", "ln1", drop_masks[0]), ("fc2", "ln2", drop_masks[1])):
        x = _lin(x, P, fc)
        x = torch.relu(F.layer_norm(x, (x.shape[-1],), P[f"{ln}.weight"], P[f"{ln}.bias"], 1e-5))
        if training and mask is not None:
            x = x * mask.to(x.dtype) / (1.0 - p_drop)
    weight = torch.softmax(_lin(x, P, "head_pi") / temp, dim=-1)
    raw = _lin(x, P, "head_mu").reshape(-1, max_K, 2)
    unit = raw / raw.norm(dim=-1, keepdim=True).clamp_min(1e-4)        # F.normalize(eps=1e-4)
    c, s = unit[..., 0], unit[..., 1]
    small = torch.sqrt(c * c + s * s) < 1e-3
    c = torch.where(small, torch.ones_like(c), c)
    s = torch.where(small, torch.zeros_like(s), s)
    mu = torch.atan2(s, c)
    kappa = F.softplus(_lin(x, P, "head_kappa")) + 1e-6
    if kappa_max is not None:
        kappa = kappa.clamp_max(kappa_max)
    return mu, kappa, weight


# --------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------
def kl_single(mu_p, kappa_p, mu_q, kappa_q):
    """train_single_peak_vonMises_KL.py:23-28 (p = prediction, q = ground truth)."""
    i0p, i1p, i0q = torch.special.i0(kappa_p), torch.special.i1(kappa_p), torch.special.i0(kappa_q)
    a1 = torch.where(kappa_p <= 1e-6, torch.zeros_like(kappa_p), i1p / i0p)
    return torch.log(i0q) - torch.log(i0p) + kappa_p * a1 - kappa_q * a1 * torch.cos(mu_p - mu_q)


def kl_multi(mu_p, kappa_p, mu_q, kappa_q):
    """train_multi_peaks_vonMises_KL.py:38-52 (clamped kappa, wrapped angle)."""
    kp = torch.clamp(kappa_p, 1e-6, 500.0)
    kq = torch.clamp(kappa_q, 1e-6, 500.0)
    i0p, i1p, i0q = torch.special.i0(kp), torch.special.i1(kp), torch.special.i0(kq)
    d = (mu_p - mu_q + math.pi) % (2 * math.pi) - math.pi
    return torch.log(i0q / i0p) + (i1p / i0p) * (kp - kq * torch.cos(d))


def match_loss(mu, kappa, w, vm_gt, K_gt, return_assignment=False):
    """train_multi_peaks_vonMises_KL.py:54-81: K x K KL cost, Hungarian matching on the host
    (scipy.optimize.linear_sum_assignment, the reference's own dependency at :75), weighted mean."""
    from scipy.optimize import linear_sum_assignment
    B = mu.shape[0]
    losses, assigns = [], []
    for b in range(B):
        K = int(K_gt[b])
        if K <= 0:
            losses.append(mu.new_zeros(()))
            assigns.append(np.zeros((0,), np.int64))
            continue
        cost = kl_multi(mu[b, :K, None], kappa[b, :K, None], vm_gt[b, None, :K, 0].to(mu.dtype),
                        vm_gt[b, None, :K, 1].to(mu.dtype))
        cost = torch.nan_to_num(cost, nan=1e6, posinf=1e6, neginf=1e6)
        row, col = linear_sum_assignment(cost.detach().cpu().numpy())
        ws = w[b, :K][row]
        losses.append((ws * cost[row, col]).sum() / (ws.sum() + 1e-8))
        assigns.append(col.astype(np.int64))
    out = torch.stack(losses)
    return (out, assigns) if return_assignment else out


def soft_ce(logits, p_target):
    """train_8dir_KL.py:60-68."""
    return -(p_target * F.log_softmax(logits, dim=1)).sum(dim=1)


def forward_axis_to_mu(axes_text: str) -> float:
    """Ground-truth angle convention: third row of the 3x3 axis file is the forward axis f,
    projected to the x-z plane; mu = atan2(f_x, -f_z); a vanishing projection degrades to -z
    (data_process/2d_single_peak_vM_gt.py:10-41, 2d_multi_peak_MvM_gt_1.py:50-59)."""
    rows = [[float(t) for t in ln.split()] for ln in axes_text.splitlines() if ln.strip()]
    if len(rows) < 3 or len(rows[2]) < 3:
        raise ValueError("axis file needs three rows of three numbers")
    fx, fz = rows[2][0], rows[2][2]
    r = math.hypot(fx, fz)
    fx, fz = (0.0, -1.0) if r < 1e-8 else (fx / r, fz / r)
    return math.atan2(fx, -fz)


# --------------------------------------------------------------------------------------
# synthetic workload and helpers shared by tests / bench cpu_baseline
# --------------------------------------------------------------------------------------
def synthetic_clouds(B: int, N: int, seed: int = 1234):
    """Anisotropic box, random yaw about Y, GT mu = atan2(f_x, -f_z), kappa = 8
    (data_process/rotate_without_normals.py:5-15, 2d_multi_peak_MvM_gt_1.py:50-59, 2d_single_peak_vM_gt.py:8)."""
    g = torch.Generator().manual_seed(seed)
    p = (torch.rand(B, N, 3, generator=g) * 2 - 1) * torch.tensor([1.0, 0.6, 0.3])
    th = torch.rand(B, generator=g) * (2 * math.pi)
    c, s = torch.cos(th), torch.sin(th)
    R = torch.zeros(B, 3, 3)
    R[:, 0, 0], R[:, 0, 2], R[:, 1, 1], R[:, 2, 0], R[:, 2, 2] = c, s, 1.0, -s, c
    xyz = torch.einsum("bnj,bij->bni", p, R).contiguous()
    f = torch.einsum("bij,j->bi", R, torch.tensor([0.0, 0.0, -1.0]))
    mu = torch.atan2(f[:, 0], -f[:, 2])
    kappa = torch.full((B,), 8.0)
    return xyz.float(), mu.float(), kappa, f


def replay_centres(B: int, sizes=((1024, 128), (128, 32)), generator: Optional[torch.Generator] = None):
    """The CPU-generator draws of pointnet_pp_8dir.py:28 in the reference's order:
    sa1: B x randperm(N)[:128], then sa2: B x randperm(128)[:32]."""
    out = []
    for n, s in sizes:
        out.append(torch.stack([torch.randperm(n, generator=generator)[:s] for _ in range(B)]))
    return out


def cast_params(state: Dict[str, torch.Tensor], dtype, requires_grad=True) -> Dict[str, torch.Tensor]:
    P = {}
    for k, v in state.items():
        if v.is_floating_point():
            t = v.detach().clone().to(dtype)
            if requires_grad and "running_" not in k:
                t.requires_grad_(True)
            P[k] = t
        else:
            P[k] = v.clone()
    return P
