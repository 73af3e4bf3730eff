"""GPU parity, heads and losses (gate G5 of SURVEY 8d): values and analytic gradients of the HIP
kernels vs the fp64 captures of the reference's own loss functions (tests/golden/kl.npz), the
reference's run log (debug_log KAT), and the fp64 oracle for the fully connected blocks."""
import math

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _rel(got, ref):
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.abs(np.asarray(got, dtype=np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30))


@pytest.fixture(scope="module")
def ops():
    from pnpp_hip import ops
    return ops


def test_kl_single_values_and_grads(ops, golden):
    g = golden("kl.npz")
    c = _t(g["single_in"]).cuda()
    mu = c[:, 0].clone().requires_grad_(True)
    kap = c[:, 1].clone().requires_grad_(True)
    v = ops.kl_von_mises_single(mu, kap, c[:, 2].contiguous(), c[:, 3].contiguous())
    v.sum().backward()
    got = np.stack([v.detach().cpu().numpy(), mu.grad.cpu().numpy(), kap.grad.cpu().numpy()], 1).astype(np.float64)
    ref = g["single_f64"]
    # G5: <= 1e-5 * max(1, |ref|) against the fp64 evaluation of the reference formula
    assert np.all(np.abs(got - ref) <= 1e-5 * np.maximum(1.0, np.abs(ref)))
    # and the fp32 reference itself (its own fp32 i0/i1 arithmetic is 2.3e-5 abs off at KL ~ 16)
    ref32 = g["single_f32"]
    assert np.all(np.abs(got - ref32) <= 2e-4 * np.maximum(1.0, np.abs(ref32)))


def test_kl_single_beyond_reference_overflow(ops):
    """fp32 i0 overflows at kappa >= 89 in the reference (NaN); the stable form stays finite and matches fp64."""
    kap = torch.tensor([89.0, 120.0, 400.0]).cuda()
    mu = torch.tensor([0.3, -1.0, 2.0]).cuda()
    v = ops.kl_von_mises_single(mu, kap, torch.zeros(3).cuda(), torch.full((3,), 8.0).cuda()).cpu().double()
    kd, md = kap.cpu().double(), mu.cpu().double()
    a = torch.special.i1e(kd) / torch.special.i0e(kd)
    q = torch.tensor(8.0, dtype=torch.float64)
    ref = (q + torch.log(torch.special.i0e(q))) - (kd + torch.log(torch.special.i0e(kd))) + kd * a - q * a * torch.cos(md)
    assert torch.isfinite(v).all() and torch.allclose(v, ref, rtol=1e-6)


def test_vm_head_and_fused_head_kl(ops, oracle):
    g = torch.Generator().manual_seed(0)
    o = (torch.randn(64, 2, generator=g) * 3)
    o[0, 1], o[1, 1], o[2, 1] = 25.0, -30.0, 0.0
    mu_gt = (torch.rand(64, generator=g) * 2 - 1) * math.pi
    kap_gt = torch.where(torch.rand(64, generator=g) < 0.3, torch.zeros(64), torch.full((64,), 8.0))
    od = o.double().requires_grad_(True)
    mu_ref = torch.tanh(od[:, 0]) * math.pi
    kap_ref = torch.nn.functional.softplus(od[:, 1])
    lv_ref = oracle.kl_single(mu_ref, kap_ref, mu_gt.double(), kap_gt.double())
    lv_ref.sum().backward()
    # autograd path: vm_head + kl_von_mises_single
    og = o.clone().cuda().requires_grad_(True)
    mu, kap = ops.vm_head(og)
    lv = ops.kl_von_mises_single(mu, kap, mu_gt.cuda(), kap_gt.cuda())
    lv.sum().backward()
    assert _rel(mu.detach().cpu(), mu_ref.detach()) < 1e-6 and _rel(kap.detach().cpu(), kap_ref.detach()) < 1e-6
    assert np.all(np.abs(lv.detach().cpu().double().numpy() - lv_ref.detach().numpy()) <= 1e-5 * np.maximum(1, np.abs(lv_ref.detach().numpy())))
    assert np.all(np.abs(og.grad.cpu().double().numpy() - od.grad.numpy()) <= 1e-5 * np.maximum(1, np.abs(od.grad.numpy())))
    # fused single-launch path gives the same numbers
    mu2, kap2, lv2, d_o = ops.vm_head_kl_fused(o.cuda(), mu_gt.cuda(), kap_gt.cuda())
    assert torch.equal(mu2, mu.detach()) and torch.equal(kap2, kap.detach()) and torch.equal(lv2, lv.detach())
    assert torch.allclose(d_o, og.grad, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("B", [1, 32, 300])
def test_vm_head_kl_loss_op(ops, oracle, B):
    """Head + KL (+ batch mean) and their gradient in one launch, against the fp64 oracle; B=300 exercises the
    strided loop and the fixed-order reduction of the single-workgroup kernel."""
    g = torch.Generator().manual_seed(B)
    o = torch.randn(B, 2, generator=g) * 3
    mu_gt = (torch.rand(B, generator=g) * 2 - 1) * math.pi
    kap_gt = torch.where(torch.rand(B, generator=g) < 0.3, torch.zeros(B), torch.full((B,), 8.0))
    od = o.double().requires_grad_(True)
    lv_ref = oracle.kl_single(torch.tanh(od[:, 0]) * math.pi, torch.nn.functional.softplus(od[:, 1]), mu_gt.double(),
                              kap_gt.double())
    lv_ref.mean().backward()
    og = o.clone().cuda().requires_grad_(True)
    loss = ops.vm_head_kl_loss(og, mu_gt.cuda(), kap_gt.cuda(), reduction="mean")
    assert loss.shape == ()
    loss.backward()
    assert abs(float(loss) - float(lv_ref.mean())) <= 1e-5 * max(1.0, abs(float(lv_ref.mean())))
    gref = od.grad.numpy()
    assert np.all(np.abs(og.grad.cpu().double().numpy() - gref) <= 1e-5 * np.maximum(1.0 / B, np.abs(gref)))
    # per-sample form, with a non-trivial upstream gradient
    og2 = o.clone().cuda().requires_grad_(True)
    lv = ops.vm_head_kl_loss(og2, mu_gt.cuda(), kap_gt.cuda(), reduction="none")
    wts = torch.rand(B, generator=g)
    (lv * wts.cuda()).sum().backward()
    od2 = o.double().requires_grad_(True)
    (oracle.kl_single(torch.tanh(od2[:, 0]) * math.pi, torch.nn.functional.softplus(od2[:, 1]), mu_gt.double(),
                      kap_gt.double()) * wts.double()).sum().backward()
    assert np.all(np.abs(lv.detach().cpu().double().numpy() - lv_ref.detach().numpy())
                  <= 1e-5 * np.maximum(1, np.abs(lv_ref.detach().numpy())))
    assert np.all(np.abs(og2.grad.cpu().double().numpy() - od2.grad.numpy()) <= 1e-5 * np.maximum(1, np.abs(od2.grad.numpy())))
    with pytest.raises(ValueError):
        ops.vm_head_kl_loss(og, mu_gt.cuda(), kap_gt.cuda(), reduction="sum")
    with pytest.raises(ValueError):
        ops.vm_head_kl_loss(og, mu_gt[:-1].cuda() if B > 1 else torch.zeros(2).cuda(), kap_gt.cuda())


def test_match_loss_vs_reference_fp64(ops, golden):
    g = golden("kl.npz")
    mu, kap, w = (_t(g[k]).cuda().requires_grad_(True) for k in ("match_mu", "match_kappa", "match_w"))
    lv = ops.match_loss(mu, kap, w, _t(g["match_vm"]).cuda(), _t(g["match_K"]).cuda())
    ref = g["match_f64_loss"].astype(np.float64)
    assert np.all(np.abs(lv.detach().cpu().double().numpy() - ref) <= 1e-5 * np.maximum(1, np.abs(ref)))
    lv.sum().backward()
    got = np.stack([mu.grad.cpu().numpy(), kap.grad.cpu().numpy(), w.grad.cpu().numpy()], 0).astype(np.float64)
    refg = g["match_f64_grads"]
    assert np.all(np.abs(got - refg) <= 1e-5 * np.maximum(1, np.abs(refg)))
    assert np.all(lv.detach().cpu().numpy()[g["match_K"] == 0] == 0)


def test_match_loss_debug_log_known_answers(ops, golden):
    """The reference's own training log: recompute every sampled block on the GPU."""
    g = golden("debug_log_kat.npz")
    for K in (1, 2, 4):
        blk = _t(g[f"K{K}"]).float()                        # (n, 7, K)
        n = blk.shape[0]
        pad = lambda x: torch.nn.functional.pad(x, (0, 4 - K))
        mu, kap, w = pad(blk[:, 0]), pad(blk[:, 1]), pad(blk[:, 2])
        vm = torch.zeros(n, 4, 3)
        vm[:, :K, 0], vm[:, :K, 1] = blk[:, 3], blk[:, 4]
        lv = ops.match_loss(mu.cuda(), kap.cuda(), w.cuda(), vm.cuda(), torch.full((n,), K).cuda()).cpu().double()
        cost, ws = blk[:, 5].double(), blk[:, 2].double()
        want = (ws * cost).sum(1) / (ws.sum(1) + 1e-8)
        assert torch.allclose(lv, want, rtol=2e-5, atol=2e-6), K


def test_mvm_head_forward_backward(ops, oracle):
    g = torch.Generator().manual_seed(5)
    B, K = 48, 4
    pi = torch.randn(B, K, generator=g)
    mr = torch.randn(B, 2 * K, generator=g) * 0.5
    mr[0] = 0.0                                             # degenerate direction -> fallback (1,0), zero gradient
    mr[1, :2] = torch.tensor([3e-5, -2e-5])                 # below the normalize eps
    kr = torch.randn(B, K, generator=g) * 3
    kr[2, 0] = 200.0                                        # clamp_max(80) active
    pd, md, kd = (t.double().requires_grad_(True) for t in (pi, mr, kr))
    w_ref = torch.softmax(pd / 0.7, -1)
    raw = md.reshape(B, K, 2)
    unit = raw / raw.norm(dim=-1, keepdim=True).clamp_min(1e-4)
    c, s = unit[..., 0], unit[..., 1]
    small = torch.sqrt(c * c + s * s) < 1e-3
    mu_ref = torch.atan2(torch.where(small, torch.zeros_like(s), s), torch.where(small, torch.ones_like(c), c))
    kap_ref = (torch.nn.functional.softplus(kd) + 1e-6).clamp_max(80.0)
    gm, gk, gw = (torch.randn(B, K, generator=g).double() for _ in range(3))
    (mu_ref * gm + kap_ref * gk + w_ref * gw).sum().backward()
    pg, mg, kg = (t.clone().cuda().requires_grad_(True) for t in (pi, mr, kr))
    mu, kap, w = ops.mvm_head(pg, mg, kg, 0.7, 80.0)
    (mu * gm.float().cuda() + kap * gk.float().cuda() + w * gw.float().cuda()).sum().backward()
    for got, ref in ((mu, mu_ref), (kap, kap_ref), (w, w_ref), (pg.grad, pd.grad), (mg.grad, md.grad), (kg.grad, kd.grad)):
        ref = ref.detach().numpy()
        assert np.all(np.abs(got.detach().cpu().double().numpy() - ref) <= 2e-5 * np.maximum(1, np.abs(ref)))


def test_soft_ce(ops, golden):
    g = golden("kl.npz")
    lg = _t(g["ce_logits"]).cuda().requires_grad_(True)
    v = ops.soft_ce(lg, _t(g["ce_p"]).cuda())
    v.sum().backward()
    assert np.allclose(v.detach().cpu().numpy(), g["ce_loss"], rtol=1e-5, atol=1e-6)
    assert np.allclose(lg.grad.cpu().numpy(), g["ce_grad"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("kind", ["bn", "ln", "none"])
@pytest.mark.parametrize("M,K,N", [(32, 1024, 512), (8, 256, 16), (5, 512, 256), (100, 512, 256), (64, 1024, 512), (300, 260, 132)])
def test_fc_block(ops, oracle, kind, M, K, N):
    torch.manual_seed(M * 7 + N)
    lin = nn.Linear(K, N)
    norm = {"bn": nn.BatchNorm1d(N), "ln": nn.LayerNorm(N), "none": None}[kind]
    if norm is not None:
        with torch.no_grad():
            norm.weight.uniform_(0.5, 1.5)
            norm.bias.uniform_(-0.5, 0.5)
    x = torch.randn(M, K)
    mask = (torch.rand(M, N) < 0.6).to(torch.uint8)
    gy = torch.randn(M, N)
    drop = nn.Dropout(0.4)
    # fp64 reference
    x64 = x.double().requires_grad_(True)
    W, b = lin.weight.detach().double().requires_grad_(True), lin.bias.detach().double().requires_grad_(True)
    z = x64 @ W.t() + b
    nw = nb = None
    if kind == "bn":
        nw, nb = norm.weight.detach().double().requires_grad_(True), norm.bias.detach().double().requires_grad_(True)
        z = (z - z.mean(0)) / torch.sqrt(z.var(0, unbiased=False) + 1e-5) * nw + nb
    elif kind == "ln":
        nw, nb = norm.weight.detach().double().requires_grad_(True), norm.bias.detach().double().requires_grad_(True)
        z = torch.nn.functional.layer_norm(z, (N,), nw, nb, 1e-5)
    y_ref = torch.relu(z) * mask.double() / 0.6
    (y_ref * gy.double()).sum().backward()
    # HIP
    lin, norm = lin.cuda(), (norm.cuda() if norm is not None else None)
    xg = x.cuda().requires_grad_(True)
    y = ops.fc_block(xg, lin, norm, relu=True, dropout=drop, training=True, mask=mask.cuda())
    y.backward(gy.cuda())
    assert _rel(y.detach().cpu(), y_ref.detach()) < 1e-5
    assert _rel(xg.grad.cpu(), x64.grad) < 2e-5
    assert _rel(lin.weight.grad.cpu(), W.grad) < 2e-5
    if kind == "bn":
        assert float(lin.bias.grad.abs().max()) == 0.0 and float(b.grad.abs().max()) < 1e-9
        assert int(norm.num_batches_tracked) == 1
    else:
        assert _rel(lin.bias.grad.cpu(), b.grad) < 2e-5
    if norm is not None:
        assert _rel(norm.weight.grad.cpu(), nw.grad) < 2e-5 and _rel(norm.bias.grad.cpu(), nb.grad) < 2e-5


def test_fc_block_eval_and_plain(ops):
    torch.manual_seed(3)
    lin, bn = nn.Linear(256, 64).cuda(), nn.BatchNorm1d(64).cuda()
    with torch.no_grad():
        bn.running_mean.uniform_(-1, 1)
        bn.running_var.uniform_(0.5, 2)
    x = torch.randn(16, 256).cuda()
    bn.eval()
    y = ops.fc_block(x, lin, bn, relu=True, dropout=nn.Dropout(0.5), training=False)
    ref = torch.relu(torch.nn.functional.batch_norm(lin(x).double().cpu(), bn.running_mean.double().cpu(),
                                                    bn.running_var.double().cpu(), bn.weight.double().cpu(),
                                                    bn.bias.double().cpu(), False, 0.1, 1e-5))
    assert _rel(y.detach().cpu(), ref.detach()) < 1e-5
    y2 = ops.fc_block(x, lin, training=False)
    assert _rel(y2.detach().cpu(), lin(x).detach().double().cpu()) < 1e-5


def test_fc_block_draws_its_dropout_mask_in_the_kernel(ops):
    """Training, BatchNorm1d, batch <= 32: the keep-mask comes from the GEMM epilogue's own counter-based generator
    (no RNG launch); it is Bernoulli(1-p), changes from call to call, and the backward pass uses exactly that mask."""
    torch.manual_seed(3)
    M, K, N, p = 32, 256, 512, 0.5
    lin, bn, drop = nn.Linear(K, N).cuda(), nn.BatchNorm1d(N).cuda(), nn.Dropout(p)
    x = torch.randn(M, K, device="cuda", requires_grad=True)
    y = ops.fc_block(x, lin, bn, relu=False, dropout=drop, training=True)
    y2 = ops.fc_block(x.detach(), lin, bn, relu=False, dropout=drop, training=True)
    keep, keep2 = (y != 0), (y2 != 0)                          # bn(z) is never exactly 0, so zeros are dropped elements
    rate = float(keep.float().mean())
    assert abs(rate - (1 - p)) < 5 * math.sqrt(p * (1 - p) / (M * N)), rate
    assert float((keep != keep2).float().mean()) > 0.3          # an independent draw on the second call
    assert float(keep.float().mean(0).min()) > 0.1 and float(keep.float().mean(1).min()) > 0.35
    # same function as BatchNorm1d (train) followed by this mask / (1 - p), forward and backward, in float64
    up = torch.randn(M, N, device="cuda")
    lin.zero_grad(), bn.zero_grad()
    (y * up).sum().backward()
    xd = x.detach().cpu().double().requires_grad_(True)
    wd, bd = lin.weight.detach().cpu().double().requires_grad_(True), lin.bias.detach().cpu().double().requires_grad_(True)
    gd, hd = bn.weight.detach().cpu().double().requires_grad_(True), bn.bias.detach().cpu().double().requires_grad_(True)
    z = xd @ wd.t() + bd
    zn = (z - z.mean(0)) / torch.sqrt(z.var(0, unbiased=False) + bn.eps) * gd + hd
    ref = zn * keep.cpu().double() / (1 - p)
    (ref * up.cpu().double()).sum().backward()
    assert _rel(y.detach().cpu().numpy(), ref.detach().numpy()) < 1e-5
    assert _rel(x.grad.cpu().numpy(), xd.grad.numpy()) < 1e-4
    assert _rel(lin.weight.grad.cpu().numpy(), wd.grad.numpy()) < 1e-4
    assert _rel(bn.weight.grad.cpu().numpy(), gd.grad.numpy()) < 1e-4 and _rel(bn.bias.grad.cpu().numpy(), hd.grad.numpy()) < 1e-4


def test_vm_head_kl_loss_backward_equals_autograd(ops):
    """loss + backward seed in one call: same loss and the same upstream gradients as loss.backward()."""
    g = torch.Generator().manual_seed(9)
    B = 32
    lin = nn.Linear(16, 2).cuda()
    x = torch.randn(B, 16, generator=g).cuda()
    mu_gt = ((torch.rand(B, generator=g) * 2 - 1) * math.pi).cuda()
    kap_gt = torch.full((B,), 8.0).cuda()
    loss_a = ops.vm_head_kl_loss(lin(x), mu_gt, kap_gt, reduction="mean")
    loss_a.backward()
    ga, gb = lin.weight.grad.clone(), lin.bias.grad.clone()
    lin.zero_grad()
    loss_b = ops.vm_head_kl_loss_backward(lin(x), mu_gt, kap_gt)
    assert not loss_b.requires_grad and float(loss_b) == float(loss_a)
    assert torch.equal(lin.weight.grad, ga) and torch.equal(lin.bias.grad, gb)


@pytest.mark.parametrize("B,K", [(32, 256), (7, 64), (300, 260)])
def test_vm_fc_head_kl_loss_backward_equals_the_unfused_ops(ops, B, K):
    """fc3 + head + KL + mean + backward in one launch vs Linear (fc_block) -> vm_head_kl_loss_backward and vs float64
    autograd of the reference's expressions (pointnet_pp_vonMises.py:35-37, train_single_peak_vonMises_KL.py:82-84)."""
    torch.manual_seed(B)
    lin = nn.Linear(K, 2).cuda()
    x = torch.randn(B, K, device="cuda")
    mu_gt = (torch.rand(B, device="cuda") * 2 - 1) * 3.1
    kappa_gt = torch.rand(B, device="cuda") * 30 + 0.5
    xa = x.clone().requires_grad_(True)
    la = ops.vm_fc_head_kl_loss_backward(xa, lin, mu_gt, kappa_gt)
    ga = (lin.weight.grad.clone(), lin.bias.grad.clone(), xa.grad.clone())
    lin.zero_grad(set_to_none=True)
    xb = x.clone().requires_grad_(True)
    lb = ops.vm_head_kl_loss_backward(ops.fc_block(xb, lin, training=True), mu_gt, kappa_gt)
    gb = (lin.weight.grad.clone(), lin.bias.grad.clone(), xb.grad.clone())
    assert abs(float(la) - float(lb)) <= 2e-6 * max(1.0, abs(float(lb)))
    for a, b in zip(ga, gb):
        assert _rel(a.cpu(), b.cpu()) < 2e-5
    # float64 reference
    x64 = x.double().cpu().requires_grad_(True)
    W, b = lin.weight.detach().double().cpu().requires_grad_(True), lin.bias.detach().double().cpu().requires_grad_(True)
    o = x64 @ W.t() + b
    mu, kappa = torch.tanh(o[:, 0]) * np.pi, torch.nn.functional.softplus(o[:, 1])
    kq = kappa_gt.double().cpu()
    i0 = lambda k: torch.special.i0e(k) * torch.exp(k)
    a1 = torch.special.i1e(kappa) / torch.special.i0e(kappa)
    loss = (torch.log(i0(kq)) - torch.log(i0(kappa)) + a1 * (kappa - kq * torch.cos(mu - mu_gt.double().cpu()))).mean()
    loss.backward()
    assert abs(float(la) - float(loss)) <= 1e-5 * max(1.0, abs(float(loss)))
    assert _rel(ga[0].cpu(), W.grad) < 2e-5 and _rel(ga[1].cpu(), b.grad) < 2e-5 and _rel(ga[2].cpu(), x64.grad) < 2e-5


@pytest.mark.parametrize("B,K,maxK", [(32, 256, 4), (7, 64, 4), (24, 128, 8), (40, 256, 4), (16, 64, 5)])
def test_mvm_heads_match_loss_backward_equals_the_unfused_ops(ops, oracle, B, K, maxK):
    """The multi-peak step's tail in one launch (three output heads, head activations, match_loss, mean, their backward:
    models/pointnet_pp_mvM.py:91-125, train_multi_peaks_vonMises_KL.py:54-81, :229-234) against the separate launches it fuses and
    against float64 autograd of the oracle's expressions.  (40 x 256 and max_K = 5 do not fit / are not instantiated: the op takes
    the separate launches itself.)"""
    torch.manual_seed(B + K)
    heads = [nn.Linear(K, n).cuda() for n in (maxK, 2 * maxK, maxK)]
    x = torch.randn(B, K, device="cuda") * 0.5
    g = torch.Generator().manual_seed(B)
    Kgt = torch.randint(0, maxK + 1, (B,), generator=g)
    vm = torch.zeros(B, maxK, 3)
    vm[:, :, 0] = (torch.rand(B, maxK, generator=g) * 2 - 1) * 3.1
    vm[:, :, 1] = torch.rand(B, maxK, generator=g) * 20 + 0.5
    vm[:, :, 2] = 1.0 / maxK
    temp, kmax = 0.7, 80.0
    xa = x.clone().requires_grad_(True)
    la, mu_a, kap_a, w_a = ops.mvm_heads_match_loss_backward(xa, *heads, vm.cuda(), Kgt.cuda(), temp, kmax, outputs=True)
    ga = [p.grad.clone() for h in heads for p in (h.weight, h.bias)] + [xa.grad.clone()]
    for h in heads:
        h.zero_grad(set_to_none=True)
    xb = x.clone().requires_grad_(True)
    mu_b, kap_b, w_b = ops.mvm_head(*[ops.fc_block(xb, h, training=True) for h in heads], temp, kmax)
    lb = ops.match_loss(mu_b, kap_b, w_b, vm.cuda(), Kgt.cuda()).mean()
    lb.backward()
    gb = [p.grad.clone() for h in heads for p in (h.weight, h.bias)] + [xb.grad.clone()]
    assert not la.requires_grad and abs(float(la) - float(lb)) <= 2e-6 * max(1.0, abs(float(lb)))
    for a, b in ((mu_a, mu_b), (kap_a, kap_b), (w_a, w_b)):
        assert _rel(a.cpu().numpy(), b.detach().cpu().numpy()) < 2e-6
    for a, b in zip(ga, gb):
        assert _rel(a.cpu().numpy(), b.cpu().numpy()) < 2e-5
    # float64: the oracle's head + match_loss on the same raw outputs
    x64 = x.double().cpu().requires_grad_(True)
    P = {n: (h.weight.detach().double().cpu().requires_grad_(True), h.bias.detach().double().cpu().requires_grad_(True))
         for n, h in zip(("pi", "mu", "kappa"), heads)}
    raw = {n: x64 @ W.t() + b for n, (W, b) in P.items()}
    weight = torch.softmax(raw["pi"] / temp, -1)
    v = raw["mu"].view(B, maxK, 2)
    u = v / v.norm(dim=-1, keepdim=True).clamp_min(1e-4)
    mu64 = torch.atan2(u[..., 1], u[..., 0])
    kap64 = (torch.nn.functional.softplus(raw["kappa"]) + 1e-6).clamp_max(kmax)
    l64 = oracle.match_loss(mu64, kap64, weight, vm.double(), Kgt).mean()
    l64.backward()
    assert abs(float(la) - float(l64)) <= 1e-5 * max(1.0, abs(float(l64)))
    ref = [t.grad for n in ("pi", "mu", "kappa") for t in P[n]] + [x64.grad]
    for a, r in zip(ga, ref):
        assert _rel(a.cpu().numpy(), r.numpy()) < 5e-5


def test_layernorm_block_draws_its_dropout_mask_in_the_kernel(ops):
    """Training, LayerNorm head block (models/pointnet_pp_mvM.py:82-83, dropout p = 0.4 twice): the keep-mask comes from the
    LayerNorm pass's own counter-based generator (no bernoulli_ launch, graph-replayable); Bernoulli(1 - p), a fresh draw per call,
    and forward / backward are exactly the block with that mask given."""
    torch.manual_seed(5)
    M, K, N, p = 32, 1024, 512, 0.4
    lin, ln, drop = nn.Linear(K, N).cuda(), nn.LayerNorm(N).cuda(), nn.Dropout(p)
    x = torch.randn(M, K, device="cuda", requires_grad=True)
    y = ops.fc_block(x, lin, ln, relu=False, dropout=drop, training=True)
    y2 = ops.fc_block(x.detach(), lin, ln, relu=False, dropout=drop, training=True)
    keep, keep2 = (y != 0), (y2 != 0)
    rate = float(keep.float().mean())
    assert abs(rate - (1 - p)) < 5 * math.sqrt(p * (1 - p) / (M * N)), rate
    assert float((keep != keep2).float().mean()) > 0.3
    assert float(keep.float().mean(0).min()) > 0.2 and float(keep.float().mean(1).min()) > 0.45
    up = torch.randn(M, N, device="cuda")
    (y * up).sum().backward()
    g1 = (x.grad.clone(), lin.weight.grad.clone(), ln.weight.grad.clone(), ln.bias.grad.clone())
    x.grad = None
    lin.zero_grad(), ln.zero_grad()
    y3 = ops.fc_block(x, lin, ln, relu=False, dropout=drop, training=True, mask=keep.to(torch.uint8))
    assert torch.equal(y3, y)
    (y3 * up).sum().backward()
    for a, b in zip(g1, (x.grad, lin.weight.grad, ln.weight.grad, ln.bias.grad)):
        assert torch.equal(a, b)
