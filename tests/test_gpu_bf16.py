"""GPU: the opt-in bf16-operand throughput mode of the grouped layers (csrc/gemm_bf16_kernels.hip; BASELINE.json configs[1]
"bf16", SURVEY 8d "bf16 mode reported with its own tolerance").  The float32 path is the parity path; what is pinned here:
  * the mode really runs the bf16 kernels (launch tags), only for the large dense GEMMs, and switching it off restores
    bit-identical float32 results;
  * its results are the float32 results up to bfloat16 operand rounding -- a wrong operand layout (row permutation,
    k order of the register-fed dW operand, swizzle) would be an O(1) error, rounding is 1e-3 ... 1e-2;
  * the measured end-to-end tolerance against the fp64 oracle at the metric configuration (printed; bounds are loose
    multiples of what was measured on MI355X)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture()
def bf16_mode():
    from pnpp_hip import ops
    assert ops.get_matmul_precision() == "f32"
    yield lambda on: ops.set_matmul_precision("bf16" if on else "f32")
    ops.set_matmul_precision("f32")


def _tags(fn):
    from pnpp_hip import _lib
    lib = _lib.lib()
    torch.cuda.synchronize()
    lib.pnpp_profile_enable(1)
    out = fn()
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 16)
    lib.pnpp_profile_report(buf, len(buf))
    lib.pnpp_profile_enable(0)
    return out, [ln.split("\t")[0] for ln in buf.value.decode().splitlines()]


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


@pytest.mark.parametrize("level", ["sa1", "sa2"])
def test_bf16_set_abstraction_equals_float32_up_to_operand_rounding(bf16_mode, level):
    from models.pointnet_pp_8dir import PointNetSetAbstraction
    torch.manual_seed(3)
    B = 8
    if level == "sa1":
        sa = PointNetSetAbstraction(128, 32, 0, [64, 64, 128]).cuda().train()
        xyz, pts = torch.rand(B, 1024, 3).cuda(), None
    else:
        sa = PointNetSetAbstraction(32, 32, 128, [128, 128, 256]).cuda().train()
        xyz, pts = torch.rand(B, 128, 3).cuda(), torch.randn(B, 128, 128).cuda().requires_grad_(True)
    g = torch.Generator().manual_seed(1)
    N, S = xyz.shape[1], sa.npoint
    centres = torch.stack([torch.randperm(N, generator=g)[:S] for _ in range(B)]).cuda()
    up = torch.randn(B, S, sa.convs[-1].out_channels, generator=g).cuda()

    def run():
        sa.zero_grad(set_to_none=True)
        if pts is not None:
            pts.grad = None
        _, out = sa(xyz, pts, centres)
        (out * up).sum().backward()
        grads = {n: p.grad.clone() for n, p in sa.named_parameters()}
        if pts is not None:
            grads["points"] = pts.grad.clone()
        return out.detach().clone(), grads

    bf16_mode(False)
    (o32, g32), t32 = _tags(run)
    bf16_mode(True)
    (o16, g16), t16 = _tags(run)
    bf16_mode(False)
    (o32b, g32b), _ = _tags(run)
    assert not any("gemm_wsb" in t for t in t32) and any("gemm_wsb" in t for t in t16), t16
    n16 = [t for t in t16 if "gemm_wsb" in t]
    assert any(",dW>" in t for t in n16) and any("E1>" in t for t in n16), n16      # forward and fused backward both ran
    assert torch.equal(o32, o32b) and all(torch.equal(g32[k], g32b[k]) for k in g32)   # the switch leaves no residue
    e_out = _rel(o16, o32)
    worst = max(((k, _rel(g16[k], g32[k])) for k in g32 if float(g32[k].abs().max()) > 0), key=lambda kv: kv[1])
    print(f"\n[{level}] bf16 vs f32: output relL2 {e_out:.2e}, worst gradient tensor {worst[0]} {worst[1]:.2e}; kernels: {len(n16)}")
    errs = {k: _rel(g16[k], g32[k]) for k in g32 if float(g32[k].abs().max()) > 0}
    print("   per tensor: " + ", ".join(f"{k} {v:.1e}" for k, v in errs.items()))
    assert 1e-6 < e_out < 2e-2, e_out
    for k in g32:
        if float(g32[k].abs().max()) == 0:           # structurally zero gradients stay exact zeros
            assert float(g16[k].abs().max()) == 0.0, k
            continue
        # operand rounding is 4e-3 per factor and grows through the cancelling sums of the BatchNorm-backward chain (largest
        # on layer 0, last in the chain); a layout error would be O(1) -- and is excluded exactly by the test below
        assert errs[k] < 0.3, (k, errs[k])


@pytest.mark.parametrize("level", ["sa1", "sa2"])
def test_bf16_products_are_exact_on_bf16_representable_data(bf16_mode, level):
    """Layout check with exact data (the guide's advice for register-fed MFMA operands): in eval mode BatchNorm is the
    identity here (running mean 0, variance 1 - eps, weight 1, bias 0), coordinates are multiples of 1/8 and the weights
    sparse with entries in {-1, 0, 1}, so every MFMA operand of every layer -- forward activations, the BatchNorm-backward
    operand dZ, relu(bn(z)) fed to the dW product from registers -- is a small dyadic number that bfloat16 holds exactly
    and every sum is exact in float32.  The bf16 kernels must then reproduce the float32 kernels BIT FOR BIT: any wrong
    row permutation, k order, swizzle or transposed-image slot shows up as a mismatch."""
    from models.pointnet_pp_8dir import PointNetSetAbstraction
    g = torch.Generator().manual_seed(7)
    B = 8
    if level == "sa1":
        sa = PointNetSetAbstraction(128, 32, 0, [64, 64, 128])
        N, D = 1024, 0
    else:
        sa = PointNetSetAbstraction(32, 32, 128, [128, 128, 256])
        N, D = 128, 128
    with torch.no_grad():
        for conv, bn in zip(sa.convs, sa.bns):
            w = torch.zeros_like(conv.weight)
            cout, cin = w.shape[:2]
            for o in range(cout):                      # two +-1 entries per output channel
                cols = torch.randperm(cin, generator=g)[:2]
                w[o, cols, 0, 0] = torch.tensor([1.0, -1.0])[torch.randint(0, 2, (2,), generator=g)]
            conv.weight.copy_(w)
            conv.bias.zero_()
            bn.weight.fill_(1.0), bn.bias.zero_(), bn.running_mean.zero_(), bn.running_var.fill_(1.0 - bn.eps)
    sa = sa.cuda().eval()
    xyz = (torch.randint(0, 8, (B, N, 3), generator=g).float() / 8).cuda()
    pts = torch.randint(-2, 3, (B, N, D), generator=g).float().cuda().requires_grad_(True) if D else None
    centres = torch.stack([torch.randperm(N, generator=g)[:sa.npoint] for _ in range(B)]).cuda()
    up = torch.randint(-2, 3, (B, sa.npoint, sa.convs[-1].out_channels), generator=g).float().cuda()

    def run():
        sa.zero_grad(set_to_none=True)
        if pts is not None:
            pts.grad = None
        _, out = sa(xyz, pts, centres)
        (out * up).sum().backward()
        grads = {n: p.grad.clone() for n, p in sa.named_parameters() if ".convs." in n or ".bns." in n or True}
        if pts is not None:
            grads["points"] = pts.grad.clone()
        return out.detach().clone(), grads

    bf16_mode(False)
    (o32, g32), _ = _tags(run)
    bf16_mode(True)
    (o16, g16), t16 = _tags(run)
    bf16_mode(False)
    n16 = [t for t in t16 if "gemm_wsb" in t]
    assert any(",dW>" in t for t in n16) and len(n16) >= 4, t16
    assert float(o32.abs().max()) > 0 and float(o32.abs().max()) < 256          # non-trivial and inside bf16's exact range
    assert torch.equal(o16, o32)
    for k in g32:
        assert torch.equal(g16[k], g32[k]), (k, float((g16[k] - g32[k]).abs().max()), float(g32[k].abs().max()))
    assert any(float(v.abs().max()) > 0 for k, v in g32.items() if "convs.1.weight" in k or "convs.2.weight" in k)


def test_bf16_model_tolerance_vs_fp64_oracle(bf16_mode, oracle):
    """configs[1] (B=32, N=1024), injected centres and mask: loss and flat gradient of the bf16 mode against fp64, next to the
    float32 path's own distance -- the measured tolerance of the mode (DESIGN.md quotes these numbers)."""
    from models.pointnet_pp_vonMises import PointNetPPVonMises
    from pnpp_hip import ops
    B, N = 32, 1024
    torch.manual_seed(42)
    model = PointNetPPVonMises()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    xyz, mu_gt, kappa_gt, _ = oracle.synthetic_clouds(B, N, seed=1234)
    torch.manual_seed(4242)
    centres = oracle.replay_centres(B)
    mask = (torch.rand(B, 256, generator=torch.Generator().manual_seed(8)) < 0.5).to(torch.uint8)
    P64 = oracle.cast_params(state, torch.float64)
    m64, k64 = oracle.vonmises_forward(xyz, P64, centres, mask.float(), True, None)
    l64 = oracle.kl_single(m64, k64, mu_gt.double(), kappa_gt.double()).mean()
    l64.backward()
    skip = lambda n: (".convs." in n and n.endswith("bias")) or n in ("fc1.bias", "fc2.bias")
    res = {}
    for mode in (False, True):
        bf16_mode(mode)
        m = PointNetPPVonMises()
        m.load_state_dict(state)
        m = m.cuda().train()
        mu, kappa = m(xyz.cuda(), centres=[c.cuda() for c in centres], drop_mask=mask.cuda())
        loss = ops.kl_von_mises_single(mu, kappa, mu_gt.cuda(), kappa_gt.cuda()).mean()
        loss.backward()
        num = den = 0.0
        for n, p in m.named_parameters():
            if skip(n):
                continue
            r = P64[n].grad.reshape(p.shape)
            num += float((p.grad.cpu().double() - r).pow(2).sum())
            den += float(r.pow(2).sum())
        res[mode] = (abs(loss.item() - l64.item()), (num / den) ** 0.5, float((mu.detach().cpu().double() - m64.detach()).abs().max()))
    bf16_mode(False)
    print(f"\n[vM B=32] vs fp64:  f32 path |dloss| {res[False][0]:.2e} grad relL2 {res[False][1]:.2e} |dmu| {res[False][2]:.2e}   "
          f"bf16 mode |dloss| {res[True][0]:.2e} grad relL2 {res[True][1]:.2e} |dmu| {res[True][2]:.2e}")
    assert res[False][0] <= 1e-5
    # measured on MI355X: |dloss| 8e-2 (1.5 % of the loss), flat gradient relL2 0.44, |dmu| 0.32 rad -- the train-mode
    # BatchNorm chain amplifies rounding ~100x (float32's 6e-8 becomes 4e-6 in the loss; bfloat16's 4e-3 becomes 1e-1)
    assert res[True][0] <= 5e-2 * max(1.0, abs(l64.item())) and res[True][1] <= 1.0 and res[True][2] <= 1.0, res[True]
