"""GPU parity of every set-abstraction level and head block of PointNetPPVonMises at the BASELINE shapes (configs[1]: 32 clouds of 1024
points per GPU), kernel family by kernel family, against the float64 oracle with EVERY discrete decision of the HIP path injected:
the max-pool routing (pnpp_sa_saved_argmax) and the ReLU decisions of every layer (pnpp_sa_saved_relu_mask).  With those given, the
float64 evaluation is a smooth function of rounding, no "an arg-max / ReLU flip explains it" remains, and what is left is the
arithmetic of the kernels themselves -- held to 1e-5 of each tensor's max-abs (VERDICT round 3, item 1 asked for 1e-4):

  sa1 (B 32, N 1024 -> 128 x 32, 3 -> 64 -> 64 -> 128):  rel_moments + gemm_wsf03 (gemm_wsf0: float32 MFMA) + gemm_wsf3<64> forward;
                                                          gemm_wsd3<128,32,A5> (L2), gemm_wsx + xyz0_post (L1 + L0) backward
  sa2 (128 -> 32 x 32, 131 -> 128 -> 128 -> 256):        gather_rel_stats + gemm_wsf3<128> forward;
                                                          gemm_wsd3<256,32,A5> (L2), gemm_wsd3<128,32,A4> (L1), scatter_dz (L0) backward
  (split products, the default; with PNPP_SPLIT_PRODUCTS=0: gemm_wsf / gemm_wsp / gemm_wsq / gemm_ws<...,dW> on the float32 MFMA pipe --
   tests/test_gpu_split_products.py runs both forms side by side)
  sa3 (group_all, 259 -> 256 -> 512 -> 1024):            gemm_smallm + gemm_mid3 (split products; gemm_mid: float32 MFMA) forward;
                                                         da_dw_mid + da_dw backward
  fc1 / fc2 (32 rows, BatchNorm1d + ReLU (+ dropout)):   gemm_smallm<E_BN_APPLY> forward; fc_bwd_fused backward

Inputs of each level are the float32 oracle's own activations and upstream gradients of the synthetic batch (so |mean| / std of
every pre-BN tensor is what the network really produces); reference being restated: models/pointnet_pp_8dir.py:21-43 and
its autograd backward, models/pointnet_pp_vonMises.py:31-35.
"""
import math

import pytest
import torch

from conftest import relmax as _relmax, routed_level

pytestmark = pytest.mark.gpu

B, N = 32, 1024
GATE = 1e-5          # of the tensor's max-abs, every output and every parameter gradient: SURVEY 8d's operator gate G2 -- for a whole
                     # three-layer level.  Measured (round 4, MI355X): 2e-8 ... 2.8e-6 with split products (the default), 2e-8 ... 1.5e-6 on
                     # the float32 MFMA pipe -- the wave-private / folded kernels of round 3 and the generic ones alike (PNPP_NO_WSP /
                     # PNPP_NO_WSQ / PNPP_NO_WSX A/B)


@pytest.fixture(scope="module")
def net(oracle):
    """Seeded PointNetPPVonMises + the float32 oracle's activations / upstream gradients at every level boundary."""
    from models.pointnet_pp_vonMises import PointNetPPVonMises
    torch.manual_seed(42)
    model = PointNetPPVonMises()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    xyz, mu_gt, kappa_gt, _ = oracle.synthetic_clouds(B, N, seed=1234)
    torch.manual_seed(7)
    centres = oracle.replay_centres(B)
    mask = (torch.rand(B, 256, generator=torch.Generator().manual_seed(8)) < 0.5).float()
    P = oracle.cast_params(state, torch.float32)
    l1_xyz, l1, _ = oracle.sa_forward(xyz, None, P, "sa1", centres[0], 32, False, True, None)
    l1.retain_grad()
    l2_xyz, l2, _ = oracle.sa_forward(l1_xyz, l1, P, "sa2", centres[1], 32, False, True, None)
    l2.retain_grad()
    _, l3, _ = oracle.sa_forward(l2_xyz, l2, P, "sa3", None, None, True, True, None)
    l3.retain_grad()
    f1 = oracle._bn1d(oracle._lin(l3.reshape(B, -1), P, "fc1"), P, "bn1", True, None).relu()
    f1.retain_grad()
    f2 = oracle._bn1d(oracle._lin(f1, P, "fc2"), P, "bn2", True, None).relu() * mask * 2.0
    f2.retain_grad()
    out = oracle._lin(f2, P, "fc3")
    mu, kappa = torch.tanh(out[:, 0]) * math.pi, torch.nn.functional.softplus(out[:, 1])
    oracle.kl_single(mu, kappa, mu_gt, kappa_gt).mean().backward()
    d = lambda t: t.detach().clone()
    return {"model": model, "state": state, "xyz": xyz, "centres": centres, "mask": mask,
            "l1_xyz": d(l1_xyz), "l1": d(l1), "d_l1": d(l1.grad), "l2_xyz": d(l2_xyz), "l2": d(l2), "d_l2": d(l2.grad),
            "l3": d(l3), "d_l3": d(l3.grad), "f1": d(f1), "d_f1": d(f1.grad), "d_f2": d(f2.grad)}


def _run_level(oracle, net, prefix, xyz, pts, centres, dout, group_all):
    sa = getattr(net["model"], prefix).cuda().train()
    P = oracle.cast_params({k: v for k, v in net["state"].items() if k.startswith(prefix + ".")}, torch.float64)
    res, diag = routed_level(oracle, sa, xyz, pts, centres, dout, 32, group_all, True, P=P, prefix=prefix)
    print(f"\n[{prefix} B={B}] ReLU decisions that differ from float64's own: {diag['relu_flips']} (margin {max(diag['relu_flip_margin']):.1e}); "
          f"routing gap {max(diag['route_gap']):.1e}\n    " + "\n    ".join(f"{k:20s} rel-to-max {v:.2e}" for k, v in res.items()))
    return res


def test_sa1_level_kernels(oracle, net):
    res = _run_level(oracle, net, "sa1", net["xyz"], None, net["centres"][0], net["d_l1"], False)
    assert max(res.values()) <= GATE, res


def test_sa2_level_kernels(oracle, net):
    res = _run_level(oracle, net, "sa2", net["l1_xyz"], net["l1"], net["centres"][1], net["d_l2"], False)
    assert max(res.values()) <= GATE, res


def test_sa3_level_kernels(oracle, net):
    res = _run_level(oracle, net, "sa3", net["l2_xyz"], net["l2"], None, net["d_l3"], True)
    assert max(res.values()) <= GATE, res


@pytest.mark.parametrize("block", ["fc1", "fc2"])
def test_head_block_kernels(oracle, net, block):
    """Linear -> BatchNorm1d -> ReLU (-> Dropout) at 32 rows: one forward launch, one backward launch (fc_bwd_fused_kernel).  No
    decision is injected here: 16 k pre-activations, none within rounding of zero on this batch (asserted)."""
    from pnpp_hip import ops
    m = net["model"].cuda().train()
    lin, bn = (m.fc1, m.bn1) if block == "fc1" else (m.fc2, m.bn2)
    x = net["l3"].reshape(B, -1) if block == "fc1" else net["f1"]
    dy = net["d_f1"] if block == "fc1" else net["d_f2"]
    mask = None if block == "fc1" else net["mask"]
    m.zero_grad()
    xg = x.cuda().requires_grad_(True)
    y = ops.fc_block(xg, lin, bn, relu=True, training=True, **({} if mask is None else {"dropout": m.drop, "mask": mask.to(torch.uint8).cuda()}))
    y.backward(dy.cuda())
    P = oracle.cast_params({k: v for k, v in net["state"].items() if k.startswith((block, "bn" + block[-1]))}, torch.float64)
    x64 = x.double().requires_grad_(True)
    pre = oracle._bn1d(oracle._lin(x64, P, block), P, "bn" + block[-1], True, None)
    assert float(pre.detach().abs().min() / pre.detach().abs().max()) > 1e-6      # no ReLU decision within rounding of zero
    y64 = pre.relu() * (1.0 if mask is None else mask.double() * 2.0)
    (y64 * dy.double()).sum().backward()
    res = {"out": _relmax(y, y64), "d_x": _relmax(xg.grad, x64.grad), "d_weight": _relmax(lin.weight.grad, P[block + ".weight"].grad),
           "d_gamma": _relmax(bn.weight.grad, P[f"bn{block[-1]}.weight"].grad), "d_beta": _relmax(bn.bias.grad, P[f"bn{block[-1]}.bias"].grad)}
    print(f"\n[{block} M={B}] " + ", ".join(f"{k} {v:.2e}" for k, v in res.items()))
    assert max(res.values()) <= GATE, res


def test_zz_wave_pair_polls_never_timed_out():
    """gemm_wsd3_kernel's producer / consumer waves poll LDS counters with a bound; a poll that gave up leaves a mark (and wrong numbers)."""
    from pnpp_hip import _lib
    torch.cuda.synchronize()
    assert _lib.lib().pnpp_debug_wsd3_timeouts() == 0
