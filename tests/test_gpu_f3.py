"""GPU parity of the other set-abstraction models and their losses: PointNetPP, PointNetPPFwd,
PointNetPPXYZ, PointNetPPXYZ_Schedmit on the HIP kernels vs the fp64 oracle (pinned to the reference's own fp64 runs
by tests/test_oracle_golden.py::test_f3_models_fp64_match_reference_fp64) and vs the reference capture itself."""
import importlib
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

F3_MODELS = {"pp": ("models.pointnet_pp", "PointNetPP"), "fwd": ("models.pointnet_pp_Fwd", "PointNetPPFwd"),
             "xyz": ("models.Pointnet_pp_xyz", "PointNetPPXYZ"), "sch": ("models.Pointnet_pp_xyz_Schedmit", "PointNetPPXYZ_Schedmit")}
ZERO_GRAD = lambda n: (".convs." in n and n.endswith("bias")) or n in ("fc1.bias", "fc2.bias")


@pytest.fixture(scope="module")
def ops():
    from pnpp_hip import ops as o
    return o


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _close(got, ref, tol):
    got, ref = got.detach().cpu().double().numpy(), ref.detach().cpu().double().numpy()
    return np.all(np.abs(got - ref) <= tol * np.maximum(1.0, np.abs(ref)))


# ---------------------------------------------------------------------------------------------- operators
def test_l2_normalize_forward_backward(ops, oracle):
    g = torch.Generator().manual_seed(0)
    for M, C in ((1, 3), (37, 3), (300, 8), (5, 64)):
        x = torch.randn(M, C, generator=g)
        if M > 4:
            x[1] = 0.0                       # ||x|| <= eps: y = 0, dx = dy / eps
            x[2] = x[2] * 1e-20              # denormal-small row, still below eps
            x[3] = x[3] * 1e4
        up = torch.randn(M, C, generator=g)
        xd = x.double().requires_grad_(True)
        (oracle.l2_normalize(xd) * up.double()).sum().backward()
        xg = x.clone().cuda().requires_grad_(True)
        y = ops.l2_normalize(xg)
        (y * up.cuda()).sum().backward()
        assert _close(y, oracle.l2_normalize(x.double()), 1e-6)
        ref = xd.grad
        got = xg.grad.cpu().double()
        assert torch.all((got - ref).abs() <= 1e-5 * ref.abs().clamp_min(1.0)), (M, C)
    with pytest.raises(ValueError):
        ops.l2_normalize(torch.zeros(4, 65, device="cuda"))
    with pytest.raises(ValueError):
        ops.l2_normalize(torch.zeros(4, device="cuda"))


def test_mse_and_orth_losses(ops, oracle):
    g = torch.Generator().manual_seed(1)
    for B, C in ((1, 3), (32, 3), (700, 8)):
        p, t = torch.randn(B, C, generator=g), torch.randn(B, C, generator=g)
        pd = p.double().requires_grad_(True)
        (3.0 * oracle.mse(pd, t.double())).backward()
        pg = p.clone().cuda().requires_grad_(True)
        loss = ops.mse_loss(pg, t.cuda())
        (3.0 * loss).backward()
        assert loss.shape == () and abs(loss.item() - oracle.mse(p.double(), t.double()).item()) <= 1e-6 * max(1.0, loss.item())
        assert torch.all((pg.grad.cpu().double() - pd.grad).abs() <= 1e-6 * pd.grad.abs().clamp_min(1e-3))
        a, b = torch.randn(B, C, generator=g), torch.randn(B, C, generator=g)
        ad, bd = a.double().requires_grad_(True), b.double().requires_grad_(True)
        ref = (ad * bd).sum(dim=1).pow(2).mean()
        ref.backward()
        ag, bg = a.clone().cuda().requires_grad_(True), b.clone().cuda().requires_grad_(True)
        got = ops.orth_loss(ag, bg)
        got.backward()
        assert abs(float(got) - float(ref)) <= 1e-6 * max(1.0, float(ref))
        assert _close(ag.grad, ad.grad, 1e-6) and _close(bg.grad, bd.grad, 1e-6)
    with pytest.raises(ValueError):
        ops.mse_loss(torch.zeros(4, 3, device="cuda"), torch.zeros(4, 2, device="cuda"))


def test_proj_probs_forward_backward(ops, oracle):
    from models.pointnet_pp_8dir import DIRS_8
    g = torch.Generator().manual_seed(2)
    B = 200
    v = torch.randn(B, 3, generator=g)
    v[0] = torch.tensor([0.0, 1.0, 0.0])     # orthogonal to every horizontal direction: all sims 0, clamped denominator
    v[1] = torch.tensor([0.0, -2.0, 0.0])
    v[2] = torch.tensor([0.0, 0.0, -3.0])    # exactly DIRS_8[0] after normalisation
    v[3] = 0.0                                # zero vector
    up = torch.randn(B, 8, generator=g)
    vd = v.double().requires_grad_(True)
    pref = oracle.proj_probs(vd, DIRS_8.double())
    (pref * up.double()).sum().backward()
    vg = v.clone().cuda().requires_grad_(True)
    p = ops.proj_probs(vg, DIRS_8.cuda())
    (p * up.cuda()).sum().backward()
    assert _close(p, pref, 1e-6)
    assert float(p[0].abs().max()) == 0.0 and float(p[3].abs().max()) == 0.0
    ok = torch.ones(B, dtype=torch.bool)
    ok[:4] = False                            # kinks (sims exactly 0 / zero vector): sub-gradient conventions differ
    ref, got = vd.grad[ok], vg.grad.cpu().double()[ok]
    assert torch.all((got - ref).abs() <= 1e-5 * ref.abs().clamp_min(1.0))
    assert torch.isfinite(vg.grad[:3]).all()


# ---------------------------------------------------------------------------------------------- models
def _hip_loss(kind, model, xyz, centres, mask, tgt):
    import losses
    from pnpp_hip import ops
    res = model(xyz, centres=centres, drop_mask=mask)
    if kind == "pp":
        return (res,), ops.mse_loss(res, tgt["fwd"])
    if kind == "fwd":
        probs = losses.proj_probs(res)
        return (res, probs), ops.mse_loss(probs, tgt["prob8"])
    ga, gb = (tgt["side"], tgt["up"]) if kind == "xyz" else (tgt["up"], tgt["fwd"])
    return res, losses.axis_pair_loss(res[0], res[1], ga, gb, 0.1)


@pytest.mark.parametrize("kind", ["pp", "fwd", "xyz", "sch"])
def test_f3_model_vs_reference_fp64_capture(oracle, golden, kind):
    """B=8, the reference's randperm draws and dropout mask injected: outputs, loss and gradient samples of the HIP
    path against what the reference produced in float64 (bounds as for the von-Mises model at B=8)."""
    g = golden("f3.npz")
    mod, cls = F3_MODELS[kind]
    torch.manual_seed(42)
    model = getattr(importlib.import_module(mod), cls)().cuda().train()
    xyz, _, _, _ = oracle.synthetic_clouds(8, 1024, seed=1234)
    centres = [_t(g["centres1"].astype(np.int64)).cuda(), _t(g["centres2"].astype(np.int64)).cuda()]
    tgt = {k: _t(g[k]).cuda() for k in ("fwd", "up", "side", "prob8")}
    for variant, mask in (("nodrop", None), ("mask", _t(g["drop_mask"]).cuda())):
        model.zero_grad()
        model.drop.p = 0.0 if mask is None else 0.5      # "nodrop": the reference ran with drop = Identity
        outs, loss = _hip_loss(kind, model, xyz.cuda(), centres, mask, tgt)
        loss.backward()
        tag = f"{kind}_f64_{variant}"
        names = {"pp": ("out",), "fwd": ("out", "probs")}.get(kind, ("out_a", "out_b"))
        for o, nm in zip(outs, names):
            assert np.abs(o.detach().cpu().double().numpy() - g[f"{tag}.{nm}"]).max() < 1e-4, (tag, nm)
        assert abs(float(loss) - float(g[f"{tag}.loss"])) <= 1e-4, tag
        worst = 0.0
        for n, p in model.named_parameters():
            if ZERO_GRAD(n):
                assert float(p.grad.abs().max()) == 0.0, n
                continue
            pos, ref, norm = g[f"{tag}.gp.{n}"], g[f"{tag}.gs.{n}"], g[f"{tag}.gn.{n}"][0]
            got = p.grad.detach().cpu().double().flatten()[pos].numpy()
            worst = max(worst, float(np.abs(got - ref).max() / max(norm / math.sqrt(p.numel()), 1e-12)))
            gn = float(p.grad.detach().double().norm())
            head = n.startswith(("fc", "bn", "head"))
            assert abs(gn - norm) <= (1e-3 if head else 2e-2) * max(norm, 1e-12), (tag, n, gn, norm)
        print(f"\n[{tag}] loss {float(loss):.7f} ref {float(g[f'{tag}.loss']):.7f} worst sampled grad err {worst:.2e}")


@pytest.mark.parametrize("kind", ["fwd", "sch"])
def test_f3_model_b32_loss_vs_fp64_oracle(oracle, kind):
    """G3 at the metric batch for two of the heads: |loss_hip - loss_fp64| <= 1e-5, head gradients <= 1e-4."""
    from models.pointnet_pp_8dir import DIRS_8
    B = 32
    mod, cls = F3_MODELS[kind]
    torch.manual_seed(42)
    model = getattr(importlib.import_module(mod), cls)()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.cuda().train()
    xyz, _, _, fwd = oracle.synthetic_clouds(B, 1024, seed=1234)
    torch.manual_seed(4242)
    centres = oracle.replay_centres(B)
    mask = (torch.rand(B, 256, generator=torch.Generator().manual_seed(8)) < 0.5).to(torch.uint8)
    up = torch.tensor([0.0, 1.0, 0.0]).expand(B, 3).contiguous()
    prob8 = torch.relu(fwd @ DIRS_8.t())
    prob8 = prob8 / prob8.sum(1, keepdim=True)
    tgt = {"fwd": fwd.cuda(), "up": up.cuda(), "prob8": prob8.cuda()}
    _, loss = _hip_loss(kind, model, xyz.cuda(), [c.cuda() for c in centres], mask.cuda(), tgt)
    loss.backward()
    P = oracle.cast_params(state, torch.float64)
    if kind == "fwd":
        out = oracle.fwd_forward(xyz, P, centres, mask.float(), True, None)
        loss64 = oracle.mse(oracle.proj_probs(out, DIRS_8.double()), prob8.double())
    else:
        va, vb = oracle.axes_forward(xyz, P, centres, ("head_y", "head_z"), mask.float(), True, None)
        loss64 = oracle.axis_pair_loss(va, vb, up.double(), fwd.double(), 0.1)
    loss64.backward()
    print(f"\n[{kind} B=32] loss hip {float(loss):.8f} fp64 {float(loss64):.8f}")
    assert abs(float(loss) - float(loss64)) <= 1e-5
    for n, p in model.named_parameters():
        if n.startswith(("fc3", "head_", "fc2.weight", "bn2")):
            ref = P[n].grad.reshape(p.shape)
            err = float((p.grad.detach().cpu().double() - ref).norm() / ref.norm())
            assert err <= 1e-4, (n, err)
