"""GPU parity end to end (gates G3/G4 of SURVEY 8d): the drop-in models on the HIP kernels vs the
fp64 oracle (itself pinned to the fp64 run of the reference, tests/test_oracle_golden.py), with the
reference's randperm centre draws and dropout mask injected.

  G3  |loss_hip - loss_fp64| <= 1e-5
  G4  flat-gradient relative L2 error vs fp64 <= 3e-3 when no discrete decision (max-pool route, ReLU) differs from float64's own,
      and <= ROUTED_GATE against float64 handed the HIP path's decisions, always
"""
import math

import numpy as np
import pytest
import torch

from conftest import FLIP_MARGIN, ROUTE_GAP, ROUTED_GATE, tap_to_routing

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.asarray(a))


ZERO_GRAD = lambda n: (".convs." in n and n.endswith("bias")) or n in ("fc1.bias", "fc2.bias")


def _flat_err(model, P, skip):
    num = den = 0.0
    for n, p in model.named_parameters():
        if skip(n):
            continue
        ref = P[n].grad.reshape(p.shape).double()
        num += float((p.grad.detach().cpu().double() - ref).pow(2).sum())
        den += float(ref.pow(2).sum())
    return math.sqrt(num / den)


def _flat_err_oracle(P32, P64, skip):
    num = den = 0.0
    for n in P64:
        if P64[n].grad is None or skip(n):
            continue
        num += float((P32[n].grad.double() - P64[n].grad).pow(2).sum())
        den += float(P64[n].grad.pow(2).sum())
    return math.sqrt(num / den)


@pytest.mark.parametrize("B", [8, 32])
def test_vonmises_loss_and_grads(oracle, golden, B):
    from models.pointnet_pp_vonMises import PointNetPPVonMises
    from pnpp_hip import ops
    torch.manual_seed(42)
    model = PointNetPPVonMises()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.cuda().train()
    xyz, mu_gt, kappa_gt, _ = oracle.synthetic_clouds(B, 1024, seed=1234)
    torch.manual_seed(4242)
    centres = oracle.replay_centres(B)
    mask = (torch.rand(B, 256, generator=torch.Generator().manual_seed(8)) < 0.5).to(torch.uint8)

    ops.sa_tap = []
    try:
        mu, kappa = model(xyz.cuda(), centres=[c.cuda() for c in centres], drop_mask=mask.cuda())
        routing = tap_to_routing(ops.sa_tap)
    finally:
        ops.sa_tap = None
    lv = ops.kl_von_mises_single(mu, kappa, mu_gt.cuda(), kappa_gt.cuda())
    loss = lv.mean()
    loss.backward()

    # float64 handed every discrete decision of the HIP path's backbone (max-pool routing, ReLU decisions): smooth in rounding
    P64r, diag = oracle.cast_params(state, torch.float64), {}
    mu64r, kap64r = oracle.vonmises_forward(xyz, P64r, centres, mask.float(), True, None, routing=routing, diag=diag)
    oracle.kl_single(mu64r, kap64r, mu_gt.double(), kappa_gt.double()).mean().backward()
    differ = sum(diag["relu_flips"]) + sum(diag["route_flips"])
    e_routed = _flat_err(model, P64r, ZERO_GRAD)
    assert max(diag["route_gap"]) <= ROUTE_GAP and max(diag["relu_flip_margin"]) <= FLIP_MARGIN, diag

    P64 = oracle.cast_params(state, torch.float64)
    st = oracle.BNState()
    mu64, kap64 = oracle.vonmises_forward(xyz, P64, centres, mask.float(), True, st)
    loss64 = oracle.kl_single(mu64, kap64, mu_gt.double(), kappa_gt.double()).mean()
    loss64.backward()
    P32 = oracle.cast_params(state, torch.float32)
    mu32, kap32 = oracle.vonmises_forward(xyz, P32, centres, mask.float(), True, None)
    loss32 = oracle.kl_single(mu32, kap32, mu_gt, kappa_gt).mean()
    loss32.backward()

    d_hip, d_cpu = abs(loss.item() - loss64.item()), abs(loss32.item() - loss64.item())
    e_hip, e_cpu = _flat_err(model, P64, ZERO_GRAD), _flat_err_oracle(P32, P64, ZERO_GRAD)
    print(f"\n[B={B}] loss hip {loss.item():.7f} fp64 {loss64.item():.7f} cpu-fp32 {loss32.item():.7f} | "
          f"|d| hip {d_hip:.2e} cpu32 {d_cpu:.2e} | grad relL2 hip {e_hip:.2e} cpu32 {e_cpu:.2e}")
    # G3 is defined at the metric batch (B=32): 1e-5.  BatchNorm1d over 8 samples is worse conditioned
    # ("same three at B=4"), there the bound is the CPU fp32 path's own distance, 1e-4.
    assert d_hip <= (1e-5 if B >= 32 else 1e-4)
    assert float((mu.detach().cpu().double() - mu64.detach()).abs().max()) < 1e-4
    # G4.  The flat gradient of the backbone is NOT a smooth function of rounding: max-over-nsample routes each pooled gradient to ONE
    # row and ReLU passes or blocks each element, and wherever a float32 evaluation sits within rounding of such a tie it may decide the
    # other way than float64 does -- an O(1) change of that element (a max-pool route: up to 8e-3 of the flat L2 norm; a ReLU decision:
    # 5e-4 ... 1e-3 of its tensor's).  The differing decisions are COUNTED (diag: float64 evaluated with the HIP path's decisions
    # injected): with none, G4's 3e-3 holds on float64's own decisions; otherwise 1e-2 -- and in every case the gradient agrees with
    # the decision-injected float64 to ROUTED_GATE, which is the gate that bounds the kernels' arithmetic.  e_cpu is a diagnostic only.
    print(f"[B={B}] {differ} backbone decisions differ from float64's own ({sum(diag['relu_flips'])} ReLU, {sum(diag['route_flips'])} max-pool); "
          f"flat gradient relL2 vs decision-injected fp64 {e_routed:.2e}")
    assert e_routed <= ROUTED_GATE * (1 if B >= 32 else 2), e_routed       # measured 9.3e-6 (B = 32) / 1.5e-5 (B = 8: BatchNorm1d over 8 samples is worse conditioned)
    assert e_hip <= (3e-3 if differ == 0 else 1e-2), (e_hip, e_cpu, differ)
    for n in ("fc1.weight", "fc2.weight", "fc3.weight", "fc3.bias", "bn1.weight", "bn2.bias"):
        p = dict(model.named_parameters())[n]
        ref = P64[n].grad.reshape(p.shape)
        err = float((p.grad.detach().cpu().double() - ref).norm() / ref.norm())
        assert err <= (1e-4 if B >= 32 else 1e-3), (n, err)
    # running statistics after the step (state_dict contract)
    sd = model.state_dict()
    for name, (rm, rv) in st.updates.items():
        assert torch.allclose(sd[name + ".running_mean"].cpu().double(), rm, rtol=1e-4, atol=1e-6), name
        assert torch.allclose(sd[name + ".running_var"].cpu().double(), rv, rtol=1e-4, atol=1e-7), name
    # structurally zero gradients are exact zeros here (noise in any fp32 autograd implementation)
    for n, p in model.named_parameters():
        if ZERO_GRAD(n):
            assert float(p.grad.abs().max()) == 0.0, n


def test_vonmises_matches_reference_capture(oracle, golden):
    """Against the numbers the reference itself produced (B=8; fp64 run: tight, fp32 run: its own conditioning)."""
    from models.pointnet_pp_vonMises import PointNetPPVonMises
    from pnpp_hip import ops
    g = golden("e2e.npz")
    torch.manual_seed(42)
    model = PointNetPPVonMises().cuda().train()
    xyz, mu_gt, kappa_gt, _ = oracle.synthetic_clouds(8, 1024, seed=1234)
    centres = [_t(g["centres1"].astype(np.int64)).cuda(), _t(g["centres2"].astype(np.int64)).cuda()]
    for variant, mask in (("nodrop", torch.ones(8, 256, dtype=torch.uint8)), ("mask", _t(g["drop_mask"]))):
        model.zero_grad()
        if variant == "nodrop":
            model.drop.p = 0.0
            mu, kappa = model(xyz.cuda(), centres=centres)
        else:
            model.drop.p = 0.5
            mu, kappa = model(xyz.cuda(), centres=centres, drop_mask=mask.cuda())
        loss = ops.kl_von_mises_single(mu, kappa, mu_gt.cuda(), kappa_gt.cuda()).mean()
        # the reference's fp64 run subtracts centre coordinates in double; ours in float32 like its fp32 run: ~1e-7 apart
        assert abs(loss.item() - float(g[f"vm_f64_{variant}.loss"])) <= 1e-4      # B=8 conditioning, see above
        assert abs(loss.item() - float(g[f"vm_f32_{variant}.loss"])) <= 3e-3
        assert np.abs(mu.detach().cpu().numpy() - g[f"vm_f64_{variant}.mu"]).max() < 1e-4


def test_mvm_and_dir8_models(oracle, golden):
    from models.pointnet_pp_mvM import PointNetPPMvM
    from models.pointnet_pp_8dir import PointNetPP8Dir
    from pnpp_hip import ops
    g = golden("e2e.npz")
    xyz, _, _, _ = oracle.synthetic_clouds(8, 1024, seed=1234)
    centres = [_t(g["centres1"].astype(np.int64)).cuda(), _t(g["centres2"].astype(np.int64)).cuda()]
    # multi-peak
    torch.manual_seed(42)
    m = PointNetPPMvM()
    torch.manual_seed(7)
    with torch.no_grad():
        m.head_pi.weight.normal_(0, 0.05)
        m.head_mu.weight.normal_(0, 0.05)
        m.head_mu.bias.normal_(0, 0.05)
    m = m.cuda().train()
    m.drop.p = 0.0
    mu, kappa, w = m(xyz.cuda(), centres=centres)                       # (B,N,3) input form
    mu2, _, _ = m(xyz.transpose(1, 2).contiguous().cuda(), centres=centres)   # (B,3,N) input form
    assert torch.equal(mu, mu2)
    for got, key, tol in ((mu, "mu", 2e-4), (kappa, "kappa", 2e-4), (w, "w", 2e-5)):
        assert np.abs(got.detach().cpu().numpy() - g[f"mvm_f64.{key}"]).max() < tol, key
    lv = ops.match_loss(mu, kappa, w, _t(g["mvm_vm_gt"]).cuda(), _t(g["mvm_K"]).cuda())
    assert np.abs(lv.detach().cpu().numpy() - g["mvm_f64.loss_vec"]).max() < 2e-4
    lv.mean().backward()
    for name in ("head_kappa.weight", "fc2.weight", "sa3.convs.2.weight", "sa1.convs.0.weight"):
        pos, ref = g[f"mvm_f64.gp.{name}"], g[f"mvm_f64.gs.{name}"]
        p = dict(m.named_parameters())[name]
        got = p.grad.detach().cpu().double().flatten()[pos].numpy()
        scale = g[f"mvm_f64.gn.{name}"][0] / math.sqrt(p.numel())
        assert np.abs(got - ref).max() <= 2e-2 * scale + 1e-9, name
        assert abs(float(p.grad.double().norm()) - g[f"mvm_f64.gn.{name}"][0]) <= 5e-3 * g[f"mvm_f64.gn.{name}"][0], name
    # 8 directions
    torch.manual_seed(42)
    d = PointNetPP8Dir().cuda().train()
    d.drop.p = 0.0
    logits = d(xyz.cuda(), centres=centres)
    assert np.abs(logits.detach().cpu().numpy() - g["dir8_f64.logits"]).max() < 2e-4
    lv = ops.soft_ce(logits, _t(g["dir8_prob"]).cuda())
    assert np.abs(lv.detach().cpu().numpy() - g["dir8_f64.loss_vec"]).max() < 2e-4
    lv.mean().backward()
    p = d.fc3.weight
    assert abs(float(p.grad.double().norm()) - g["dir8_f64.gn.fc3.weight"][0]) <= 5e-3 * g["dir8_f64.gn.fc3.weight"][0]


def test_samplers_and_eval(oracle):
    """Default sampler replays the CPU generator exactly like the reference; the device sampler needs no host RNG."""
    from models.pointnet_pp_vonMises import PointNetPPVonMises
    from models.pointnet_pp_8dir import PointNetSetAbstraction
    torch.manual_seed(42)
    model = PointNetPPVonMises().cuda().train()
    model.drop.p = 0.0
    xyz, _, _, _ = oracle.synthetic_clouds(4, 1024, seed=9)
    torch.manual_seed(77)
    mu_a, _ = model(xyz.cuda())
    torch.manual_seed(77)
    centres = oracle.replay_centres(4)
    mu_b, _ = model(xyz.cuda(), centres=[c.cuda() for c in centres])
    assert torch.equal(mu_a, mu_b)
    st = torch.get_rng_state()
    old = PointNetSetAbstraction.sampler
    try:
        PointNetSetAbstraction.sampler = "device"
        mu_c, kap_c = model(xyz.cuda())
        assert torch.equal(st, torch.get_rng_state())          # host generator untouched
        assert torch.isfinite(mu_c).all() and torch.isfinite(kap_c).all()
    finally:
        PointNetSetAbstraction.sampler = old
    model.eval()
    with torch.no_grad():
        mu_e, kap_e = model(xyz.cuda(), centres=[c.cuda() for c in centres])
    assert torch.isfinite(mu_e).all() and (kap_e >= 0).all()


def test_flat_adam_matches_torch_adam(oracle):
    """FlatAdam (one fused launch, gradients written straight into the flat buffer) vs torch.optim.Adam
    (train_single_peak_vonMises_KL.py:70,80-85) with identical centres and masks.

    Two steps only: Adam's update m/(sqrt(v)+eps) is +-lr for ANY gradient well above eps=1e-8, including the
    structurally-zero ones (sa*.bns.2.bias, conv biases' neighbours) that are pure fp32 noise, so last-bit
    differences between two correct Adam implementations are amplified to O(lr) from the third step on."""
    import copy
    from models.pointnet_pp_vonMises import PointNetPPVonMises
    from pnpp_hip import ops, optim
    torch.manual_seed(42)
    m1 = PointNetPPVonMises().cuda().train()
    m2 = copy.deepcopy(m1)
    m3 = copy.deepcopy(m1)
    o1 = optim.FlatAdam(m1.parameters(), lr=1e-3)
    o2 = torch.optim.Adam(m2.parameters(), lr=1e-3)
    o3 = optim.FlatAdam(m3.parameters(), lr=1e-3, fused_grads=False)
    xyz, mu_gt, kappa_gt, _ = oracle.synthetic_clouds(8, 1024, seed=3)
    xyz, mu_gt, kappa_gt = xyz.cuda(), mu_gt.cuda(), kappa_gt.cuda()
    for it in range(2):
        torch.manual_seed(100 + it)
        centres = [c.cuda() for c in oracle.replay_centres(8)]
        mask = (torch.rand(8, 256) < 0.5).to(torch.uint8).cuda()
        losses = []
        for m, o in ((m1, o1), (m2, o2), (m3, o3)):
            o.zero_grad()
            mu, kappa = m(xyz, centres=centres, drop_mask=mask)
            loss = ops.kl_von_mises_single(mu, kappa, mu_gt, kappa_gt).mean()
            loss.backward()
            losses.append(loss.item())
        # step 0 starts from identical parameters: identical numbers; afterwards the two Adam implementations
        # differ in the last bit of the parameters, which the B=8 network turns into a few ulp of the loss
        assert abs(losses[0] - losses[1]) <= (0.0 if it == 0 else 1e-4) and losses[0] == losses[2], (it, losses)
        for (n, a), b, c in zip(m1.named_parameters(), m2.parameters(), m3.parameters()):
            assert torch.equal(a.grad, c.grad), n                 # fused gradient sinks change nothing
            if it == 0:
                assert torch.equal(a.grad, b.grad), n
        if it == 1:   # gradient norm helper (train_multi_peaks_vonMises_KL.py:235)
            ref = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m1.parameters()))   # the helper's own gradients
            assert abs(float(o1.grad_norm()) - float(ref)) < 1e-5 * float(ref)
        o1.step(), o2.step(), o3.step(zero_grad=True)           # o3: the update also clears the gradients it consumed
        assert float(o3.flat_g.abs().max()) == 0.0 and float(o1.flat_g.abs().max()) > 0.0
        for (n, a), b, c in zip(m1.named_parameters(), m2.parameters(), m3.parameters()):
            assert torch.equal(a, c), n
            if it == 0:                                           # later steps: see the docstring
                assert torch.allclose(a, b, rtol=0, atol=2e-6), (it, n)
            else:
                assert float((a - b).abs().max()) <= 2.5e-3, (it, n)   # never more than the two +-lr steps apart
    for a, b in zip(m1.buffers(), m2.buffers()):
        assert torch.allclose(a.double(), b.double(), rtol=1e-5, atol=1e-6)


def test_hipgraph_step_equals_eager(oracle):
    """zero_grad + forward + loss + backward captured into one hipGraph: replays reproduce the eager launches
    bit for bit, and because the sampler's counter lives in device memory every replay draws fresh centres."""
    import copy
    from models.pointnet_pp_vonMises import PointNetPPVonMises
    from models.pointnet_pp_8dir import PointNetSetAbstraction
    from pnpp_hip import ops, optim, sampling
    from pnpp_hip.graph import GraphedStep
    old = PointNetSetAbstraction.sampler
    PointNetSetAbstraction.sampler = "device"
    try:
        torch.manual_seed(42)
        m1 = PointNetPPVonMises().cuda().train()
        m1.drop.p = 0.0                                    # dropout draws differ between capture and eager streams
        m2 = copy.deepcopy(m1)
        o1, o2 = optim.FlatAdam(m1.parameters()), optim.FlatAdam(m2.parameters())
        xyz, mu_gt, kappa_gt, _ = oracle.synthetic_clouds(8, 1024, seed=3)
        xyz, mu_gt, kappa_gt = xyz.cuda(), mu_gt.cuda(), kappa_gt.cuda()

        def loss_fn(model):
            return lambda x, m, k: ops.kl_von_mises_single(*model(x), m, k).mean()

        g = GraphedStep(o1, loss_fn(m1), [xyz, mu_gt, kappa_gt])
        for p, q in zip(m1.buffers(), m2.buffers()):       # warm-up / capture advanced m1's running statistics
            p.copy_(q)
        losses = []
        for it in range(3):
            sampling.reset(100 * it)
            l1 = float(g(xyz, mu_gt, kappa_gt))
            sampling.reset(100 * it)
            o2.zero_grad()
            l2 = loss_fn(m2)(xyz, mu_gt, kappa_gt)
            l2.backward()
            assert l1 == float(l2), (it, l1, float(l2))
            assert torch.equal(o1.flat_g, o2.flat_g)
            losses.append(l1)
        assert len(set(losses)) == 3                       # different counters -> different centres -> different losses
        a = float(g(xyz, mu_gt, kappa_gt))
        b = float(g(xyz, mu_gt, kappa_gt))
        assert a != b                                      # consecutive replays keep drawing
    finally:
        PointNetSetAbstraction.sampler = old


def test_hipgraph_step_with_captured_adam_equals_eager(oracle):
    """fused_optimizer=True: forward + loss + backward + Adam in one hipGraph, the step count in device memory and
    the gradients cleared by the update itself.  Five replays leave the parameters, both Adam moments and the loss
    history bit-identical to five eager zero_grad / backward / step() iterations."""
    import copy
    from models.pointnet_pp_vonMises import PointNetPPVonMises
    from models.pointnet_pp_8dir import PointNetSetAbstraction
    from pnpp_hip import ops, optim, sampling
    from pnpp_hip.graph import GraphedStep
    old = PointNetSetAbstraction.sampler
    PointNetSetAbstraction.sampler = "device"
    try:
        torch.manual_seed(7)
        m1 = PointNetPPVonMises().cuda().train()
        m1.drop.p = 0.0
        m2 = copy.deepcopy(m1)
        o1, o2 = optim.FlatAdam(m1.parameters(), lr=1e-3), optim.FlatAdam(m2.parameters(), lr=1e-3)
        xyz, mu_gt, kappa_gt, _ = oracle.synthetic_clouds(8, 1024, seed=4)
        xyz, mu_gt, kappa_gt = xyz.cuda(), mu_gt.cuda(), kappa_gt.cuda()

        def loss_fn(model):
            return lambda x, m, k: ops.kl_von_mises_single(*model(x), m, k).mean()

        g = GraphedStep(o1, loss_fn(m1), [xyz, mu_gt, kappa_gt], fused_optimizer=True)
        assert o1.step_count == 0 and torch.equal(o1.flat_p, o2.flat_p)    # capture and warm-up took no optimiser step
        for p, q in zip(m1.buffers(), m2.buffers()):
            p.copy_(q)
        for it in range(5):
            sampling.reset(10 * it)
            l1 = float(g(xyz, mu_gt, kappa_gt))
            sampling.reset(10 * it)
            o2.zero_grad()
            l2 = loss_fn(m2)(xyz, mu_gt, kappa_gt)
            l2.backward()
            o2.step()
            assert l1 == float(l2), (it, l1, float(l2))
            assert torch.equal(o1.flat_p, o2.flat_p), it
        assert o1.step_count == 5 and int(o1._step_state[0]) == 5 and int(o1._step_state[1]) == 0
        assert torch.equal(o1.exp_avg, o2.exp_avg) and torch.equal(o1.exp_avg_sq, o2.exp_avg_sq)
        assert float(o1.flat_g.abs().max()) == 0.0                           # cleared by the update
        o1.step_dev(zero_grad=False)                                         # eager use of the same entry point: a 6th step
        assert o1.step_count == 6 and int(o1._step_state[0]) == 6
    finally:
        PointNetSetAbstraction.sampler = old


def test_split_hipgraph_step_equals_eager(oracle):
    """The two-graph step used under data parallelism (forward + sa3/head backward | sa2/sa1 backward, the gradient
    slices handed to an all-reduce callback in between): same loss and, bit for bit, the same flat gradient as eager
    launches; the callback sees the final sa3 + head slice before the second graph has run."""
    import copy
    from models.pointnet_pp_vonMises import PointNetPPVonMises
    from models.pointnet_pp_8dir import PointNetSetAbstraction
    from pnpp_hip import ops, optim, sampling
    from pnpp_hip.graph import GraphedSplitStep
    old = PointNetSetAbstraction.sampler
    PointNetSetAbstraction.sampler = "device"
    try:
        torch.manual_seed(42)
        m1 = PointNetPPVonMises().cuda().train()
        m1.drop.p = 0.0
        m2 = copy.deepcopy(m1)
        o1, o2 = optim.FlatAdam(m1.parameters()), optim.FlatAdam(m2.parameters())
        xyz, mu_gt, kappa_gt, _ = oracle.synthetic_clouds(8, 1024, seed=3)
        xyz, mu_gt, kappa_gt = xyz.cuda(), mu_gt.cuda(), kappa_gt.cuda()

        def stage1(x, m, k):
            l1_xyz, l1_pts = m1.sa1(x, None)
            return m1.sa2(l1_xyz, l1_pts)

        def stage2(l2_xyz, l2_pts):
            _, l3 = m1.sa3(l2_xyz, l2_pts)
            f = ops.fc_block(l3.view(l3.size(0), -1), m1.fc1, m1.bn1, relu=True, training=True)
            f = ops.fc_block(f, m1.fc2, m1.bn2, relu=True, dropout=m1.drop, training=True)
            return ops.vm_head_kl_loss(ops.fc_block(f, m1.fc3, training=True), mu_gt, kappa_gt, reduction="mean")

        tail = o1.offset_of(next(m1.sa3.parameters()))
        assert 0 < tail < o1.numel and tail == sum(p.numel() for p in list(m1.sa1.parameters()) + list(m1.sa2.parameters()))
        g = GraphedSplitStep(o1, stage1, stage2, [xyz, mu_gt, kappa_gt], tail)
        for p, q in zip(m1.buffers(), m2.buffers()):
            p.copy_(q)
        seen = []

        class _Done:
            def wait(self):
                pass

        def fake_all_reduce(t):
            seen.append((t.data_ptr(), t.numel(), t.clone()))
            return _Done()

        for it in range(2):
            sampling.reset(100 * it)
            seen.clear()
            l1 = float(g(xyz, mu_gt, kappa_gt, all_reduce=fake_all_reduce))
            sampling.reset(100 * it)
            o2.zero_grad()
            l2 = ops.vm_head_kl_loss(m2.features(xyz), mu_gt, kappa_gt, reduction="mean")
            l2.backward()
            assert l1 == float(l2), (it, l1, float(l2))
            assert torch.equal(o1.flat_g, o2.flat_g)
            # first callback: the tail slice, already final (equal to the eager gradient) before graph 2 was replayed
            assert seen[0][1] == o1.numel - tail and torch.equal(seen[0][2], o2.flat_g[tail:])
            assert seen[1][1] == tail and torch.equal(seen[1][2], o2.flat_g[:tail])
    finally:
        PointNetSetAbstraction.sampler = old


def test_hipgraph_replays_are_bitwise_reproducible(oracle):
    """Race detection by determinism: the captured training step replayed 100 times from the same sampler
    and dropout counters produces, every time, bit for bit the flat gradient and the loss of the eager launches -- every
    cross-workgroup reduction of the step is a fixed-order sum of per-worker partials, so any ordering bug, missing barrier
    or stale read between the ~60 kernels of a replay shows up as a differing bit."""
    import copy
    from models.pointnet_pp_vonMises import PointNetPPVonMises
    from pnpp_hip import ops, optim, sampling
    from pnpp_hip.graph import GraphedStep
    torch.manual_seed(42)
    m1 = PointNetPPVonMises(sampler="device").cuda().train()
    m2 = copy.deepcopy(m1)
    o1, o2 = optim.FlatAdam(m1.parameters()), optim.FlatAdam(m2.parameters())
    xyz, mu_gt, kappa_gt, _ = oracle.synthetic_clouds(32, 1024, seed=3)
    xyz, mu_gt, kappa_gt = xyz.cuda(), mu_gt.cuda(), kappa_gt.cuda()
    g = GraphedStep(o1, lambda x, m, k: ops.vm_head_kl_loss_backward(m1.features(x), m, k), [xyz, mu_gt, kappa_gt], adopt_inputs=True)
    snap = sampling.snapshot()                      # centre-sampler and in-kernel dropout counters
    o2.zero_grad()
    ref_loss = ops.vm_head_kl_loss_backward(m2.features(xyz), mu_gt, kappa_gt)
    ref_g, ref_l = o2.flat_g.clone(), float(ref_loss)
    for it in range(100):
        sampling.restore(snap)
        loss = g(xyz, mu_gt, kappa_gt)
        assert float(loss) == ref_l, (it, float(loss), ref_l)
        assert torch.equal(o1.flat_g, ref_g), it


def test_centres_drawn_one_step_ahead_by_the_tail_launch(oracle):
    """sampling.CentreRing + pnpp_vm_fc_head_kl_step_sample (bench.py's default step): the draws of step t+1 ride in step t's
    single-workgroup tail launch.  Same counter sequence, same kernels' arithmetic: every step's loss and the parameters after
    three optimiser steps are bit for bit those of the loop that opens each step with its own sampling launch
    (models/pointnet_pp_8dir.py:28 of sa1 and sa2, drawn per step)."""
    import copy
    from models.pointnet_pp_vonMises import PointNetPPVonMises
    from pnpp_hip import ops, optim, sampling
    torch.manual_seed(42)
    m1 = PointNetPPVonMises(sampler="device").cuda().train()
    m2 = copy.deepcopy(m1)
    o1, o2 = optim.FlatAdam(m1.parameters()), optim.FlatAdam(m2.parameters())
    xyz, mu_gt, kappa_gt, _ = oracle.synthetic_clouds(8, 1024, seed=3)
    xyz, mu_gt, kappa_gt = xyz.cuda(), mu_gt.cuda(), kappa_gt.cuda()
    m1.trunk(xyz)                                        # creates the device-side counters
    snap = sampling.snapshot()
    for p, q in zip(m1.buffers(), m2.buffers()):
        p.copy_(q)

    def run(model, opt, ring):
        sampling.restore(snap)
        losses = []
        for _ in range(3):
            opt.zero_grad()
            loss = ops.vm_fc_head_kl_loss_backward(model.trunk(xyz), model.fc3, mu_gt, kappa_gt,
                                                   next_centres=ring.job() if ring is not None else None)
            opt.step()
            losses.append(float(loss))
        return losses

    la = run(m2, o2, None)                                # every step opens with its own centre draw
    ring = sampling.CentreRing(8, 1024, m1.sa1.npoint, m1.sa2.npoint, xyz.device)
    m1.use_presampled(ring)
    lb = run(m1, o1, ring)                                # primed once, then refilled by each step's tail launch
    assert la == lb, (la, lb)
    assert torch.equal(o1.flat_p, o2.flat_p)
    # after step 3 the ring already holds the centres step 4 would draw
    sampling.restore(snap)
    want = None
    for _ in range(4):
        want = sampling.device_random_centres_pair(8, 1024, m1.sa1.npoint, m1.sa1.npoint, m1.sa2.npoint, xyz.device)
    assert torch.equal(ring.c1, want[0]) and torch.equal(ring.c2, want[1])
