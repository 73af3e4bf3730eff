"""GPU parity of the point-transformer forward path: kernels vs float64 restatements, and the drop-in
model vs the numbers the reference model itself produced (tests/golden/pt.npz, eval mode and train mode with p = 0)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _pt_model():
    from models.point_transformer import PointTransformer
    torch.manual_seed(42)
    m = PointTransformer()
    torch.manual_seed(5)
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.02 * torch.randn_like(p))
    return m


def test_attention_forward_vs_float64():
    from pnpp_hip import transformer as T
    g = torch.Generator().manual_seed(0)
    for B, N, H in ((2, 256, 4), (1, 128, 1), (3, 384, 2)):
        E = 16 * H
        qkv = torch.randn(B, N, 3 * E, generator=g) * 1.5
        qkv[0, :, :E] *= 3.0                                   # a cloud with peaked softmax rows
        out, lse = T.attention(qkv.cuda(), H, want_lse=True)
        q, k, v = (t.double().reshape(B, N, H, 16).transpose(1, 2) for t in qkv.split(E, dim=-1))
        s = (q * 0.25) @ k.transpose(-1, -2)
        ref = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, N, E)
        assert float((out.cpu().double() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max())), (B, N, H)
        assert float((lse.cpu().double() - torch.logsumexp(s, dim=-1)).abs().max()) <= 2e-5
        # backward (scores recomputed in two kernels: dQ, then dK / dV) vs float64 autograd
        up = torch.randn(B, N, E, generator=g)
        qd = qkv.double().requires_grad_(True)
        q2, k2, v2 = (t.reshape(B, N, H, 16).transpose(1, 2) for t in qd.split(E, dim=-1))
        ((torch.softmax((q2 * 0.25) @ k2.transpose(-1, -2), dim=-1) @ v2).transpose(1, 2).reshape(B, N, E) * up.double()).sum().backward()
        qg = qkv.clone().cuda().requires_grad_(True)
        (T.attention(qg, H) * up.cuda()).sum().backward()
        err = float((qg.grad.cpu().double() - qd.grad).abs().max()) / max(1.0, float(qd.grad.abs().max()))
        assert err <= 2e-5, (B, N, H, err)
    with pytest.raises(ValueError):
        T.attention(torch.zeros(1, 100, 192, device="cuda"), 4)          # N not a multiple of 128
    with pytest.raises(ValueError):
        T.attention(torch.zeros(1, 128, 96, device="cuda"), 4)           # head dimension 8


def test_layernorm_mean_and_small_linear_vs_float64():
    from pnpp_hip import transformer as T
    g = torch.Generator().manual_seed(1)
    x, r = torch.randn(1000, 64, generator=g) * 3 + 1, torch.randn(1000, 64, generator=g)
    ln = torch.nn.LayerNorm(64)
    with torch.no_grad():
        ln.weight.normal_(1.0, 0.2, generator=g), ln.bias.normal_(0.0, 0.2, generator=g)
    ref = torch.nn.functional.layer_norm((x + r).double(), (64,), ln.weight.double(), ln.bias.double(), 1e-5)
    got = T.add_layernorm(x.cuda(), r.cuda(), ln.cuda())
    assert float((got.detach().cpu().double() - ref.detach()).abs().max()) <= 2e-6 * max(1.0, float(ref.detach().abs().max()))
    # backward: du for both addends, d weight, d bias
    up = torch.randn(1000, 64, generator=g)
    xd, rd = x.double().requires_grad_(True), r.double().requires_grad_(True)
    wd, bd = ln.weight.detach().cpu().double().requires_grad_(True), ln.bias.detach().cpu().double().requires_grad_(True)
    (torch.nn.functional.layer_norm(xd + rd, (64,), wd, bd, 1e-5) * up.double()).sum().backward()
    xg, rg = x.clone().cuda().requires_grad_(True), r.clone().cuda().requires_grad_(True)
    ln.zero_grad()
    (T.add_layernorm(xg, rg, ln) * up.cuda()).sum().backward()
    for got_g, ref_g in ((xg.grad, xd.grad), (rg.grad, rd.grad), (ln.weight.grad, wd.grad), (ln.bias.grad, bd.grad)):
        assert float((got_g.cpu().double() - ref_g).abs().max()) <= 1e-5 * max(1.0, float(ref_g.abs().max()))
    xs = torch.randn(3, 777, 64, generator=g)
    assert float((T.mean_points(xs.cuda()).cpu().double() - xs.double().mean(1)).abs().max()) <= 1e-6
    lin = torch.nn.Linear(3, 64)
    pts = torch.randn(5000, 3, generator=g)
    ref = pts.double() @ lin.weight.double().t() + lin.bias.double()
    lin = lin.cuda()
    y = T.linear_smallk(pts.cuda(), lin)
    assert float((y.detach().cpu().double() - ref.detach()).abs().max()) <= 1e-6
    upl = torch.randn(5000, 64, generator=g)
    (y * upl.cuda()).sum().backward()
    assert float((lin.weight.grad.cpu().double() - upl.double().t() @ pts.double()).abs().max()) <= 1e-5 * 5000 ** 0.5
    assert float((lin.bias.grad.cpu().double() - upl.double().sum(0)).abs().max()) <= 1e-5 * 5000 ** 0.5
    xm = xs.clone().cuda().requires_grad_(True)
    upm = torch.randn(3, 64, generator=g)
    (T.mean_points(xm) * upm.cuda()).sum().backward()
    assert float((xm.grad.cpu().double() - (upm.double() / 777)[:, None, :].expand(3, 777, 64)).abs().max()) <= 1e-9


def test_point_transformer_forward_matches_reference_capture(golden):
    g = golden("pt.npz")
    model = _pt_model().cuda().eval()
    out = model(_t(g["xyz"]).cuda()).detach().cpu().double().numpy()
    ref64, ref32 = g["pt_f64.eval_out"], g["pt_f32.eval_out"]
    d64 = np.abs(out - ref64).max()
    print(f"\nHIP vs reference fp64: {d64:.2e}; reference fp32 vs its own fp64: {np.abs(ref32 - ref64).max():.2e}")
    assert d64 <= 2e-5 * max(1.0, np.abs(ref64).max())
    assert np.abs(out - g["pt_f64.train_out"]).max() <= 2e-5 * max(1.0, np.abs(ref64).max())   # train mode, p = 0


def test_point_transformer_training_step_matches_reference_capture(golden):
    """Train mode with the dropout probabilities at 0 and the MSE harness loss: loss and every parameter gradient against
    the reference's own float64 autograd run."""
    from pnpp_hip import ops
    g = golden("pt.npz")
    model = _pt_model().cuda().train().set_dropout(0.0)
    out = model(_t(g["xyz"]).cuda())
    loss = ops.mse_loss(out, _t(g["target"]).cuda())
    loss.backward()
    assert np.abs(out.detach().cpu().double().numpy() - g["pt_f64.train_out"]).max() <= 2e-5
    assert abs(loss.item() - float(g["pt_f64.loss"])) <= 1e-5 * max(1.0, float(g["pt_f64.loss"]))
    worst = 0.0
    for n, p in model.named_parameters():
        pos, ref, norm = g[f"pt_f64.gp.{n}"], g[f"pt_f64.gs.{n}"], g[f"pt_f64.gn.{n}"][0]
        got = p.grad.detach().cpu().double().flatten()[pos].numpy()
        scale = max(norm / math.sqrt(p.numel()), 1e-12)
        worst = max(worst, float(np.abs(got - ref).max() / scale))
        gn = float(p.grad.detach().double().norm())
        assert abs(gn - norm) <= 1e-4 * max(norm, 1e-12), (n, gn, norm)
    print(f"\nloss {loss.item():.8f} ref {float(g['pt_f64.loss']):.8f}; worst sampled gradient error / rms {worst:.2e}")
    assert worst <= 1e-3


def test_point_transformer_forward_larger_cloud_vs_oracle(oracle):
    model = _pt_model()
    P64 = {k: v.double() for k, v in model.state_dict().items()}
    xyz, _, _, _ = oracle.synthetic_clouds(2, 1024, seed=3)
    ref = oracle.point_transformer_forward(xyz.double(), P64)
    out = model.cuda().eval()(xyz.cuda()).detach().cpu().double()
    assert float((out - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))


def _unpack(mask, N):
    """(B,H,N,N/32) int32 bit-packed -> dense (B,H,N,N) float64 of {0,1} (bit j of word w = column 32 w + j)."""
    m = mask.cpu().numpy().view(np.uint32)
    bits = ((m[..., None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(*m.shape[:3], N)
    return torch.from_numpy(bits.astype(np.float64))


def test_attention_dropout_masks_and_same_mask_parity():
    """Keep bits: Bernoulli(1-p) with independent rows / columns, both orientations consistent, a pure function of the
    seed; and the dropped attention (forward and backward) against float64 autograd evaluated with the SAME mask."""
    from pnpp_hip import transformer as T
    B, N, H, p = 2, 256, 4, 0.1
    mask, maskT = T.attention_dropout_mask(B, N, H, p, "cuda", seed=11, stream_id=3)
    keep = _unpack(mask, N)
    assert torch.equal(keep, _unpack(maskT, N).transpose(-1, -2))                  # maskT is the transpose of mask
    again, _ = T.attention_dropout_mask(B, N, H, p, "cuda", seed=11, stream_id=3)
    other, _ = T.attention_dropout_mask(B, N, H, p, "cuda", seed=11, stream_id=4)
    assert torch.equal(mask, again) and not torch.equal(mask, other)
    n = keep.numel()
    assert abs(float(keep.mean()) - (1 - p)) < 4 * math.sqrt(p * (1 - p) / n)
    rows, cols = keep.mean(-1), keep.mean(-2)                                     # per-row / per-column keep rates
    assert float((rows - (1 - p)).abs().max()) < 6 * math.sqrt(p * (1 - p) / N)
    assert float((cols - (1 - p)).abs().max()) < 6 * math.sqrt(p * (1 - p) / N)
    k = keep - keep.mean()
    for shift in ((0, 1), (1, 0), (1, 1)):                                         # neighbouring bits are uncorrelated
        c = float((k[..., :N - shift[0], :N - shift[1]] * k[..., shift[0]:, shift[1]:]).mean() / k.var())
        assert abs(c) < 5 / math.sqrt(n), (shift, c)
    g = torch.Generator().manual_seed(4)
    E = 16 * H
    qkv = torch.randn(B, N, 3 * E, generator=g)
    up = torch.randn(B, N, E, generator=g)
    qd = qkv.double().requires_grad_(True)
    q, kk, v = (t.reshape(B, N, H, 16).transpose(1, 2) for t in qd.split(E, dim=-1))
    w = torch.softmax((q * 0.25) @ kk.transpose(-1, -2), dim=-1) * keep / (1 - p)  # F.dropout semantics with this mask
    ref = (w @ v).transpose(1, 2).reshape(B, N, E)
    (ref * up.double()).sum().backward()
    qg = qkv.clone().cuda().requires_grad_(True)
    out = T.attention(qg, H, p=p, masks=(mask, maskT))
    (out * up.cuda()).sum().backward()
    assert float((out.detach().cpu().double() - ref.detach()).abs().max()) <= 2e-5 * max(1.0, float(ref.detach().abs().max()))
    err = float((qg.grad.cpu().double() - qd.grad).abs().max()) / max(1.0, float(qd.grad.abs().max()))
    assert err <= 2e-5, err


def test_point_transformer_trains_with_default_dropout():
    """Train mode with the constructor's dropout 0.1 everywhere: finite loss and gradients for every parameter, different
    draws on consecutive calls, eval unaffected; and the loss of a fixed batch goes down under FlatAdam."""
    from pnpp_hip import ops, optim
    import synthetic
    torch.manual_seed(0)
    model = _pt_model().cuda().train()
    xyz, _, _, fwd = synthetic.rotated_clouds(4, 256, seed=2)
    xyz, tgt = xyz.cuda(), fwd.cuda()
    a, b = model(xyz).detach(), model(xyz).detach()
    assert torch.isfinite(a).all() and not torch.equal(a, b)                        # fresh masks every call
    model.eval()
    assert torch.equal(model(xyz).detach(), model(xyz).detach())
    model.train()
    opt = optim.FlatAdam(model.parameters(), lr=1e-3)
    hist = []
    for _ in range(40):
        opt.zero_grad()
        loss = ops.mse_loss(model(xyz), tgt)
        loss.backward()
        opt.step()
        hist.append(loss.item())
    assert all(v == v for v in hist) and sum(hist[-5:]) / 5 < 0.7 * sum(hist[:5]) / 5, (hist[:5], hist[-5:])
    for n, prm in model.named_parameters():
        assert prm.grad is None or torch.isfinite(prm.grad).all(), n


def test_attention_dropout_streams_are_disjoint_and_device_counted():
    """Consecutive draws share no Philox counter word (an earlier layout made call k+1 a bit-shifted copy of call k), and the
    default stream id lives in device memory: the kernel itself advances it, so a captured step draws fresh masks on replay."""
    from pnpp_hip import transformer as T
    B, N, H, p = 1, 256, 2, 0.5
    m3, _ = T.attention_dropout_mask(B, N, H, p, "cuda", seed=11, stream_id=3)
    m4, _ = T.attention_dropout_mask(B, N, H, p, "cuda", seed=11, stream_id=4)
    a, b = m3.cpu().numpy().astype(np.uint32), m4.cpu().numpy().astype(np.uint32)
    for sh in (0, 4, 8):   # call k+1 is no shifted copy of call k: agreement of the overlapping bits stays at chance level
        agree = np.unpackbits((~((a >> sh) ^ b) & (0xFFFFFFFF >> sh)).view(np.uint8)).mean() * 32 / (32 - sh)
        assert 0.47 < agree < 0.53, (sh, agree)
    cnt = T._att_counter(torch.device("cuda"))
    before = int(cnt[0])
    d1, _ = T.attention_dropout_mask(B, N, H, p, "cuda", seed=11)
    d2, _ = T.attention_dropout_mask(B, N, H, p, "cuda", seed=11)
    assert int(cnt[0]) == before + 2 and int(cnt[1]) == 0 and not torch.equal(d1, d2)
    e1, _ = T.attention_dropout_mask(B, N, H, p, "cuda", seed=11, stream_id=before + 1)   # default id = 1 + draws so far
    assert torch.equal(d1, e1)
    # captured: the replayed launch reads the counter at run time
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        T.attention_dropout_mask(B, N, H, p, "cuda", seed=11)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        gm, _ = T.attention_dropout_mask(B, N, H, p, "cuda", seed=11)
    g.replay()
    r1 = gm.clone()
    g.replay()
    assert not torch.equal(r1, gm)


@pytest.mark.parametrize("N", [200, 1000])
def test_point_transformer_any_cloud_size(oracle, N):
    """models/point_transformer.py:15-20 takes any number of points (the reference's data has 10,000, not a multiple of the
    kernels' 128-query blocks): the drop-in pads, the attention kernels give the padding no weight, the pooling leaves it
    out -- output, loss and every parameter gradient against float64 autograd of the restatement on the UNPADDED cloud."""
    from pnpp_hip import ops
    import synthetic
    model = _pt_model()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    xyz, _, _, fwd = synthetic.rotated_clouds(3, N, seed=N)
    model = model.cuda().eval()
    with torch.no_grad():
        out_eval = model(xyz.cuda()).cpu().double()
    model.train().set_dropout(0.0)
    out = model(xyz.cuda())
    loss = ops.mse_loss(out, fwd.cuda())
    loss.backward()
    P64 = oracle.cast_params(state, torch.float64)
    ref = oracle.point_transformer_forward(xyz.double(), P64)
    l64 = ((ref - fwd.double()) ** 2).mean()
    l64.backward()
    scale = max(1.0, float(ref.detach().abs().max()))
    assert float((out_eval - ref.detach()).abs().max()) <= 2e-5 * scale
    assert float((out.detach().cpu().double() - ref.detach()).abs().max()) <= 2e-5 * scale
    assert abs(loss.item() - float(l64.detach())) <= 1e-5 * max(1.0, float(l64.detach()))
    worst = 0.0
    for n, p in model.named_parameters():
        r = P64[n].grad.reshape(p.shape)
        worst = max(worst, float((p.grad.detach().cpu().double() - r).norm() / r.norm().clamp_min(1e-30)))
    print(f"\n[PT N={N}] worst per-tensor gradient relL2 {worst:.2e}")
    assert worst <= 1e-3, worst
    # train mode with the default dropout still runs on a padded cloud
    model.set_dropout(0.1)
    ops.mse_loss(model(xyz.cuda()), fwd.cuda()).backward()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())
