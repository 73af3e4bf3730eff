"""GPU parity at the REAL per-GPU sizes of every BASELINE.json config, against the fp64 oracle
(oracle/restatement.py, itself pinned to the reference's own fp64 runs in tests/test_oracle_golden.py):

  configs[1]  PointNetPPVonMises  N=1024,  B=32          -> tests/test_gpu_e2e.py (G3/G4) + the routed gate here
  configs[2]  PointNetPPMvM + match_loss, N=1024, B=32   -> test_mvm_config2_full_batch
  configs[3]  PointNetPP8Dir     N=2048,  B=32 per GPU   -> test_dir8_config3_full_size
  configs[4]  PointTransformer   N=4096,  B=8  per GPU   -> test_point_transformer_config4_full_size
  the reference scripts' own training size N=10,000, B=16 (train_single_peak_vonMises_KL.py:18)
                                                         -> test_vonmises_reference_training_size

Random draws (centres, dropout masks) are injected so that both sides see identical inputs; neighbour sets are compared
bit-exactly (sorted) with the C oracle of the float32 distance recipe.
"""
import math

import numpy as np
import pytest
import torch

from conftest import ROUTED_GATE, tap_to_routing

pytestmark = pytest.mark.gpu

ZERO_GRAD = lambda n: (".convs." in n and n.endswith("bias")) or n in ("fc1.bias", "fc2.bias")


def _own_gate(res):
    """The flat-gradient gate on float64's OWN decisions (no injection).  Every decision on which the HIP path and float64 differ
    -- counted, not guessed: res["decisions_differ"] = ReLU decisions + max-pool routes, both float32 rounding away from a tie -- is
    an O(1) change of one element's gradient: a max-pool route is worth up to 8e-3 of the flat L2 norm, a ReLU decision 5e-4 ... 1e-3
    of its tensor's.  G4's 3e-3 when none differs; 1e-2 otherwise."""
    return 3e-3 if res["decisions_differ"] == 0 else 1e-2


def _flat(model_or_P, skip, names):
    out = []
    for n in names:
        if skip(n):
            continue
        g = model_or_P[n].grad
        out.append(g.detach().cpu().double().reshape(-1))
    return torch.cat(out)


def _rel(a, b):
    return float((a - b).norm() / b.norm())


def _knn_sets_equal(oracle, xyz, centre_idx, k):
    from pnpp_hip import ops
    new_xyz = oracle.index_points(xyz, centre_idx)
    ref = np.sort(oracle.knn_indices(new_xyz, xyz, k).numpy(), axis=-1)
    got = np.sort(ops.knn(new_xyz.cuda(), xyz.cuda(), k).cpu().numpy(), axis=-1)
    return np.array_equal(ref, got)


def _run_bn_head_model(oracle, model_cls, oracle_fwd, loss_hip, loss_ref, B, N, seed_centres, extra_skip=()):
    """Shared body: HIP model vs fp64 oracle vs CPU-fp32 oracle at (B, N) with injected centres and dropout mask."""
    torch.manual_seed(42)
    model = model_cls()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.cuda().train()
    xyz, mu_gt, kappa_gt, fwd = oracle.synthetic_clouds(B, N, seed=1234)
    torch.manual_seed(seed_centres)
    centres = oracle.replay_centres(B, sizes=((N, 128), (128, 32)))
    mask = (torch.rand(B, 256, generator=torch.Generator().manual_seed(8)) < 0.5).to(torch.uint8)

    # G1 at this size: neighbour sets of both grouped levels, bit-exact
    assert _knn_sets_equal(oracle, xyz, centres[0], 32)
    l1_xyz = oracle.index_points(xyz, centres[0])
    assert _knn_sets_equal(oracle, l1_xyz, centres[1], 32)

    from pnpp_hip import ops
    ops.sa_tap = []
    try:
        out = model(xyz.cuda(), centres=[c.cuda() for c in centres], drop_mask=mask.cuda())
        routing = tap_to_routing(ops.sa_tap)
    finally:
        ops.sa_tap = None
    loss = loss_hip(out, mu_gt, kappa_gt, fwd)
    loss.backward()

    res = {}
    # "f64r": float64 with the HIP path's neighbour order and max-pool routing injected (smooth in rounding, see
    # oracle.sa_forward); "f64" / "f32": the restatement's own routing in float64 / float32
    for tag, dt in (("f64r", torch.float64), ("f64", torch.float64), ("f32", torch.float32)):
        P = oracle.cast_params(state, dt)
        st = oracle.BNState()
        kw = dict(routing=routing, diag=res.setdefault("diag", {})) if tag == "f64r" else {}
        o = oracle_fwd(xyz, P, centres, mask.float(), True, st, **kw)
        l = loss_ref(o, mu_gt.to(dt), kappa_gt.to(dt), fwd.to(dt))
        l.backward()
        res[tag] = (P, o, l, st)
    names = [n for n, _ in model.named_parameters()]
    skip = lambda n: ZERO_GRAD(n) or n in extra_skip
    hipP = dict(model.named_parameters())
    g_hip, g64, g32 = _flat(hipP, skip, names), _flat(res["f64"][0], skip, names), _flat(res["f32"][0], skip, names)
    # the routed gate: float64 is handed EVERY discrete decision of the HIP path -- neighbour order, max-pool routing, the ReLU decisions
    # of all nine backbone layers (pnpp_sa_saved_argmax / pnpp_sa_saved_relu_mask) -- and is then a smooth function of rounding.
    # (1) every injected decision is float32 rounding away from float64's own: a routed row is a maximum up to 5e-6 relative (a float32
    # dot product of K = 512 ... 1024 terms carries ~ sqrt(K) * 6e-8 of rounding; measured <= 2.0e-6), a flipped ReLU decision sits on a
    # pre-activation that is zero up to 2e-5 of the layer's largest (measured <= 1e-7);
    # (2) given the decisions, the WHOLE flat gradient agrees with float64 to float32 accuracy
    diag = res["diag"]
    gaps = diag["route_gap"]
    e_routed = _rel(g_hip, _flat(res["f64r"][0], skip, names))
    d_routed = abs(loss.item() - res["f64r"][2].item())
    contrib = sorted(((float((hipP[n].grad.detach().cpu().double().reshape(-1) - res["f64r"][0][n].grad.reshape(-1)).pow(2).sum()), n)
                      for n in names if not skip(n)), reverse=True)[:3]
    res["decisions_differ"] = sum(diag["relu_flips"]) + sum(diag["route_flips"])
    print(f"\n[{model_cls.__name__} N={N} B={B}] routed gate: {sum(diag['relu_flips'])} ReLU decisions / {sum(diag['route_flips'])} max-pool routes "
          f"differ from float64's own (margins {max(diag['relu_flip_margin']):.1e} / {max(gaps):.1e}), |dloss| {d_routed:.2e}, "
          f"flat gradient relL2 vs routed fp64 {e_routed:.2e}; largest contributions: "
          + ", ".join(f"{n} {math.sqrt(v) / float(res['f64r'][0][n].grad.norm()):.1e}" for v, n in contrib))
    assert len(gaps) == 3 and max(gaps) <= 5e-6, gaps
    assert max(diag["relu_flip_margin"]) <= 2e-5, diag["relu_flip_margin"]
    assert d_routed <= 1e-5
    # with no discrete decision left open the flat gradient sits at float32 rounding carried through eleven normalisation layers
    # (the head's two ReLU layers, 32 x 768 decisions, are not injected).  ROUTED_GATE = 2 x the largest value measured over the five
    # full-size cases of this file
    assert e_routed <= ROUTED_GATE, e_routed
    return model, out, loss, res, (g_hip, g64, g32)


@pytest.mark.parametrize("B,N", [(32, 2048), (16, 10_000)])
def test_dir8_config3_full_size(oracle, B, N):
    """configs[3]: models/pointnet_pp_8dir.py:58-85 + train_8dir_KL.py:60-68 at N=2048, 32 clouds per GPU -- and at the
    script's own size, 10,000 points and batch 16 (train_8dir_KL.py:22-23)."""
    from models.pointnet_pp_8dir import PointNetPP8Dir, DIRS_8
    from pnpp_hip import ops
    import synthetic

    def prob(fwd):
        return synthetic.dir8_soft_labels(fwd.float(), DIRS_8)

    model, logits, loss, res, (g_hip, g64, g32) = _run_bn_head_model(
        oracle, PointNetPP8Dir, oracle.dir8_forward,
        lambda o, m, k, f: ops.soft_ce(o, prob(f).cuda()).mean(),
        lambda o, m, k, f: oracle.soft_ce(o, prob(f).to(o.dtype)).mean(), B, N, seed_centres=N)
    _, lg64, l64, st = res["f64"]
    _, _, l32, _ = res["f32"]
    d_hip, d_cpu = abs(loss.item() - l64.item()), abs(l32.item() - l64.item())
    e_hip, e_cpu, e_pair = _rel(g_hip, g64), _rel(g32, g64), _rel(g_hip, g32)
    print(f"\n[8dir N={N} B={B}] loss hip {loss.item():.7f} fp64 {l64.item():.7f} | |d| hip {d_hip:.2e} cpu32 {d_cpu:.2e} | "
          f"grad relL2 hip {e_hip:.2e} cpu32 {e_cpu:.2e} hip-vs-cpu32 {e_pair:.2e}")
    assert d_hip <= 1e-5
    assert float((logits.detach().cpu().double() - lg64.detach()).abs().max()) <= 1e-4
    # float64 on its own decisions: see _own_gate; the gate that bounds the kernels' arithmetic is the routed one inside
    # _run_bn_head_model.  e_cpu is printed as a diagnostic only and bounds nothing.
    assert e_hip <= _own_gate(res), (e_hip, e_cpu, res["decisions_differ"])
    for n in ("fc1.weight", "fc2.weight", "fc3.weight", "fc3.bias"):
        ref = res["f64"][0][n].grad
        p = dict(model.named_parameters())[n]
        assert _rel(p.grad.detach().cpu().double(), ref.reshape(p.shape)) <= 1e-4, n
    sd = model.state_dict()
    for name, (rm, rv) in st.updates.items():
        assert torch.allclose(sd[name + ".running_mean"].cpu().double(), rm, rtol=1e-4, atol=1e-6), name
        assert torch.allclose(sd[name + ".running_var"].cpu().double(), rv, rtol=1e-4, atol=1e-7), name


def test_vonmises_reference_training_size(oracle):
    """The reference scripts' own size: N=10,000 points, batch 16 (train_single_peak_vonMises_KL.py:18)."""
    from models.pointnet_pp_vonMises import PointNetPPVonMises
    from pnpp_hip import ops
    B, N = 16, 10_000
    model, (mu, kappa), loss, res, (g_hip, g64, g32) = _run_bn_head_model(
        oracle, PointNetPPVonMises, oracle.vonmises_forward,
        lambda o, m, k, f: ops.kl_von_mises_single(o[0], o[1], m.cuda(), k.cuda()).mean(),
        lambda o, m, k, f: oracle.kl_single(o[0], o[1], m, k).mean(), B, N, seed_centres=10_000)
    _, (mu64, _), l64, _ = res["f64"]
    _, _, l32, _ = res["f32"]
    d_hip, d_cpu = abs(loss.item() - l64.item()), abs(l32.item() - l64.item())
    e_hip, e_cpu = _rel(g_hip, g64), _rel(g32, g64)
    print(f"\n[vM N={N} B={B}] loss hip {loss.item():.7f} fp64 {l64.item():.7f} | |d| hip {d_hip:.2e} cpu32 {d_cpu:.2e} | "
          f"grad relL2 hip {e_hip:.2e} cpu32 {e_cpu:.2e}")
    assert d_hip <= 1e-5
    assert float((mu.detach().cpu().double() - mu64.detach()).abs().max()) < 1e-4
    assert e_hip <= _own_gate(res), (e_hip, e_cpu, res["decisions_differ"])


def test_vonmises_config1_routed_gradient_and_eval(oracle):
    """configs[1] at B=32 (N=1024): beside the fp64 gate of test_gpu_e2e.py -- which arg-max flips loosen to the CPU
    float32 path's own error -- the routed gate of _run_bn_head_model holds the WHOLE flat gradient (backbone included) to
    ROUTED_GATE of float64, unconditionally, once float64 is told which rows the max-pool took and which way every ReLU fell.  (The HIP path and the CPU float32 restatement do
    not share their routing: on the GPU box's 32 host threads the latter sits 8.7e-3 from float64, the HIP path 8.7e-4.)
    Eval mode is compared with the oracle as well: running statistics after this training step, no dropout."""
    from models.pointnet_pp_vonMises import PointNetPPVonMises
    from pnpp_hip import ops
    B, N = 32, 1024
    model, (mu, kappa), loss, res, (g_hip, g64, g32) = _run_bn_head_model(
        oracle, PointNetPPVonMises, oracle.vonmises_forward,
        lambda o, m, k, f: ops.kl_von_mises_single(o[0], o[1], m.cuda(), k.cuda()).mean(),
        lambda o, m, k, f: oracle.kl_single(o[0], o[1], m, k).mean(), B, N, seed_centres=4242)
    e_hip, e_cpu = _rel(g_hip, g64), _rel(g32, g64)
    print(f"[vM B=32] grad relL2 (own routing): hip-vs-fp64 {e_hip:.2e}, cpu32-vs-fp64 {e_cpu:.2e}")
    assert abs(loss.item() - res["f64"][2].item()) <= 1e-5
    assert e_hip <= _own_gate(res), (e_hip, e_cpu, res["decisions_differ"])
    xyz, _, _, _ = oracle.synthetic_clouds(B, N, seed=1234)
    torch.manual_seed(4242)
    centres = oracle.replay_centres(B)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    model.eval()
    with torch.no_grad():
        mu_e, kap_e = model(xyz.cuda(), centres=[c.cuda() for c in centres])
    P64e = oracle.cast_params(sd, torch.float64, requires_grad=False)
    mu_r, kap_r = oracle.vonmises_forward(xyz, P64e, centres, None, False, None)
    assert float((mu_e.cpu().double() - mu_r).abs().max()) <= 1e-4
    assert float((kap_e.cpu().double() - kap_r).abs().max()) <= 1e-4 * max(1.0, float(kap_r.abs().max()))


@pytest.mark.parametrize("B,N", [(32, 1024), (16, 10_000)])
def test_mvm_config2_full_batch(oracle, B, N):
    """configs[2]: PointNetPPMvM (models/pointnet_pp_mvM.py:30-127) + match_loss
    (train_multi_peaks_vonMises_KL.py:54-81) at N=1024, B=32 -- and at the script's own size, 10,000 points and batch 16
    (train_multi_peaks_vonMises_KL.py:25-26) -- both dropout masks injected, K_gt in {1,2,4}."""
    from models.pointnet_pp_mvM import PointNetPPMvM
    from pnpp_hip import ops
    import synthetic
    torch.manual_seed(42)
    m = PointNetPPMvM()
    torch.manual_seed(7)
    with torch.no_grad():     # the zero-initialised pi / mu heads make every angle the degenerate fallback: move off it
        m.head_pi.weight.normal_(0, 0.05)
        m.head_mu.weight.normal_(0, 0.05)
        m.head_mu.bias.normal_(0, 0.05)
    state = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.cuda().train()
    xyz, _, _, fwd = oracle.synthetic_clouds(B, N, seed=1234)
    g = torch.Generator().manual_seed(3)
    K = torch.tensor([1, 2, 4])[torch.randint(0, 3, (B,), generator=g)]
    vm_gt = synthetic.multi_peak_gt(fwd, K)
    torch.manual_seed(99)
    centres = oracle.replay_centres(B, sizes=((N, 128), (128, 32)))
    masks = [(torch.rand(B, w, generator=g) < 0.6).to(torch.uint8) for w in (512, 256)]

    ops.sa_tap = []
    try:
        mu, kappa, w = m(xyz.cuda(), centres=[c.cuda() for c in centres], drop_masks=[t.cuda() for t in masks])
        routing = tap_to_routing(ops.sa_tap)
    finally:
        ops.sa_tap = None
    lv = ops.match_loss(mu, kappa, w, vm_gt.cuda(), K.cuda())
    loss = lv.mean()
    loss.backward()

    P64 = oracle.cast_params(state, torch.float64)
    mu64, kap64, w64 = oracle.mvm_forward(xyz, P64, centres, [t.double() for t in masks], True, None)
    lv64 = oracle.match_loss(mu64, kap64, w64, vm_gt.double(), K)
    lv64.mean().backward()
    # the same with the HIP path's max-pool routing injected: the whole flat gradient, backbone included, tightly
    P64r, diag = oracle.cast_params(state, torch.float64), {}
    oracle.match_loss(*oracle.mvm_forward(xyz, P64r, centres, [t.double() for t in masks], True, None, routing=routing,
                                          diag=diag), vm_gt.double(), K).mean().backward()
    d = abs(loss.item() - lv64.mean().item())
    print(f"\n[mvM N={N} B={B}] loss hip {loss.item():.7f} fp64 {lv64.mean().item():.7f} |d| {d:.2e}")
    assert d <= 1e-5 * max(1.0, abs(lv64.mean().item()))
    assert float((lv.detach().cpu().double() - lv64.detach()).abs().max()) <= 1e-4
    assert float((w.detach().cpu().double() - w64.detach()).abs().max()) <= 2e-5
    for n in ("head_kappa.weight", "head_mu.weight", "head_pi.weight", "fc2.weight", "ln1.weight", "fc1.weight"):
        p = dict(m.named_parameters())[n]
        ref = P64[n].grad.reshape(p.shape)
        assert _rel(p.grad.detach().cpu().double(), ref) <= 2e-4, n
    names = [n for n, _ in m.named_parameters()]
    skip = lambda n: ".convs." in n and n.endswith("bias")
    e = _rel(_flat(dict(m.named_parameters()), skip, names), _flat(P64, skip, names))
    er = _rel(_flat(dict(m.named_parameters()), skip, names), _flat(P64r, skip, names))
    hipP = dict(m.named_parameters())
    contrib = sorted(((float((hipP[n].grad.detach().cpu().double().reshape(-1) - P64r[n].grad.reshape(-1)).pow(2).sum()), n)
                      for n in names if not skip(n)), reverse=True)[:4]
    differ = sum(diag["relu_flips"]) + sum(diag["route_flips"])
    print(f"[mvM N={N} B={B}] flat gradient relL2 vs fp64 {e:.2e} (float64's own decisions; {differ} differ from the HIP path's), {er:.2e} "
          f"(HIP decisions injected, margins {max(diag['relu_flip_margin']):.1e} / {max(diag['route_gap']):.2e}); largest contributions: "
          + ", ".join(f"{n} {math.sqrt(v) / float(P64r[n].grad.norm()):.1e}" for v, n in contrib))
    assert e <= (3e-3 if differ == 0 else 1e-2), (e, differ)      # see _own_gate
    assert max(diag["route_gap"]) <= 5e-6 and max(diag["relu_flip_margin"]) <= 2e-5 and er <= ROUTED_GATE, (diag, er)


def test_point_transformer_config4_full_size(oracle):
    """configs[4]: models/point_transformer.py:4-20 at N=4096, 8 clouds per GPU: eval output, and a training step
    (dropout 0, MSE harness loss on the forward axis) with every parameter gradient, against float64 autograd of the
    restatement.  The oracle runs cloud by cloud (nothing couples clouds in this model: LayerNorm is per row), so the
    N x N float64 attention of one cloud at a time is all it holds."""
    from models.point_transformer import PointTransformer
    from pnpp_hip import ops
    import synthetic
    B, N = 8, 4096
    torch.manual_seed(42)
    model = PointTransformer()
    torch.manual_seed(5)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.02 * torch.randn_like(p))
    state = {k: v.clone() for k, v in model.state_dict().items()}
    xyz, _, _, fwd = synthetic.rotated_clouds(B, N, seed=77)
    tgt = fwd.float()

    model = model.cuda().eval()
    with torch.no_grad():
        out_eval = model(xyz.cuda()).cpu().double()
    model.train().set_dropout(0.0)
    out = model(xyz.cuda())
    loss = ops.mse_loss(out, tgt.cuda())
    loss.backward()

    P64 = oracle.cast_params(state, torch.float64)
    ref_rows, loss64 = [], 0.0
    for b in range(B):                                         # per cloud: the mean over B*3 elements is a sum of cloud terms
        o = oracle.point_transformer_forward(xyz[b:b + 1].double(), P64)
        l = ((o - tgt[b:b + 1].double()) ** 2).sum() / (B * 3)
        l.backward()
        ref_rows.append(o.detach())
        loss64 += float(l.detach())
    ref = torch.cat(ref_rows)
    scale = max(1.0, float(ref.abs().max()))
    d_eval = float((out_eval - ref).abs().max())
    d_train = float((out.detach().cpu().double() - ref).abs().max())
    worst = 0.0
    for n, p in model.named_parameters():
        r = P64[n].grad.reshape(p.shape)
        worst = max(worst, _rel(p.grad.detach().cpu().double(), r))
    print(f"\n[PT N={N} B={B}] eval |d| {d_eval:.2e} train |d| {d_train:.2e} loss hip {loss.item():.8f} fp64 {loss64:.8f} "
          f"worst per-tensor gradient relL2 {worst:.2e}")
    assert d_eval <= 2e-5 * scale and d_train <= 2e-5 * scale
    assert abs(loss.item() - loss64) <= 1e-5 * max(1.0, loss64)
    assert worst <= 1e-3, worst
