"""GPU parity of BASELINE configs[0]: the drop-in `simple_pointnet_train.SimplePointNet` (per-point MLP + whole-cloud max on
the set-abstraction kernels, fused head, row-wise MSE) against the reference's own float64 run (tests/golden/simple.npz,
256 points, batch 4) and, at the script's real size (10,000 points, batch 16), against the float64 oracle with the HIP
path's max-pool routing injected (the same routed gate as tests/test_gpu_fullsize.py)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ZERO_GRAD = ("conv1.bias", "conv2.bias", "conv3.bias", "fc1.bias")


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _model():
    import simple_pointnet_train as spt
    torch.manual_seed(42)
    m = spt.SimplePointNet()
    return m, {k: v.clone() for k, v in m.state_dict().items()}


def test_mse_rows_forward_backward(oracle):
    from pnpp_hip import ops
    g = torch.Generator().manual_seed(1)
    for B, C in ((1, 3), (4, 3), (700, 8)):
        p, t, up = torch.randn(B, C, generator=g), torch.randn(B, C, generator=g), torch.randn(B, generator=g)
        pd = p.double().requires_grad_(True)
        (oracle.mse_rows(pd, t.double()) * up.double()).sum().backward()
        pg = p.clone().cuda().requires_grad_(True)
        lv = ops.mse_rows(pg, t.cuda())
        (lv * up.cuda()).sum().backward()
        ref = oracle.mse_rows(p.double(), t.double())
        assert lv.shape == (B,) and torch.all((lv.cpu().double() - ref).abs() <= 1e-6 * ref.clamp_min(1.0))
        assert torch.all((pg.grad.cpu().double() - pd.grad).abs() <= 1e-6 * pd.grad.abs().clamp_min(1e-3))
        assert abs(float(lv.detach().mean()) - float(oracle.mse(p.double(), t.double()))) <= 1e-6
    with pytest.raises(ValueError):
        ops.mse_rows(torch.zeros(4, 3, device="cuda"), torch.zeros(4, 2, device="cuda"))


def test_simple_pointnet_vs_reference_capture(oracle, golden):
    """256 points, batch 4: outputs, loss, gradients, running statistics after the step's forward and the eval-mode
    outputs that follow, against what the reference produced in float64."""
    import simple_pointnet_train as spt
    g = golden("simple.npz")
    xyz, _, _, fwd = oracle.synthetic_clouds(4, 256, seed=77)
    for variant, mask in (("nodrop", None), ("mask", _t(g["drop_mask"]).cuda())):
        model, _ = _model()
        model = model.cuda().train()
        model.dropout.p = 0.0 if mask is None else 0.3
        out = model(xyz.cuda(), drop_mask=mask)
        loss = spt.criterion(out, fwd.cuda()).mean()
        loss.backward()
        tag = f"f64_{variant}"
        assert np.abs(out.detach().cpu().double().numpy() - g[f"{tag}.out"]).max() < 2e-5, tag
        assert abs(float(loss) - float(g[f"{tag}.loss"])) <= 1e-5, tag
        worst = 0.0
        for n, p in model.named_parameters():
            if n in ZERO_GRAD:
                assert float(p.grad.abs().max()) <= 1e-6, n
                continue
            pos, ref, norm = g[f"{tag}.gp.{n}"], g[f"{tag}.gs.{n}"], g[f"{tag}.gn.{n}"][0]
            if norm < 1e-5:        # bn3.bias: removed by bn4 when a pooled channel is positive in every sample
                assert float(p.grad.abs().max()) <= 1e-5, n
                continue
            got = p.grad.detach().cpu().double().flatten()[pos].numpy()
            worst = max(worst, float(np.abs(got - ref).max() / (norm / math.sqrt(p.numel()))))
            gn = float(p.grad.detach().double().norm())
            assert abs(gn - norm) <= 1e-3 * norm, (tag, n, gn, norm)
        assert worst <= 2e-2, (tag, worst)
        print(f"\n[simple {tag}] loss {float(loss):.7f} ref {float(g[f'{tag}.loss']):.7f} worst sampled grad err {worst:.2e}")
        if mask is None:
            for k, v in model.state_dict().items():
                if "running" in k:
                    assert np.allclose(v.cpu().double().numpy(), g[f"{tag}.after.{k}"], rtol=1e-4, atol=1e-6), k
                if k.endswith("num_batches_tracked"):
                    assert int(v) == 1
            model.eval()
            with torch.no_grad():
                ev = model(xyz.cuda())
            assert np.abs(ev.cpu().double().numpy() - g[f"{tag}.eval_out"]).max() < 5e-5


@pytest.mark.parametrize("B,N", [(16, 10_000), (3, 777)])
def test_simple_pointnet_script_size_routed(oracle, B, N):
    """The script's own size (10,000 points, batch 16: the max runs over whole clouds on the split-K pooling kernels) and a
    ragged one: HIP vs the float64 oracle routed through the HIP path's arg-max."""
    from pnpp_hip import ops
    import simple_pointnet_train as spt
    model, state = _model()
    model = model.cuda().train()
    xyz, _, _, fwd = oracle.synthetic_clouds(B, N, seed=5)
    mask = (torch.rand(B, 128, generator=torch.Generator().manual_seed(3)) < 0.7).to(torch.uint8)
    ops.sa_tap = []
    try:
        out = model(xyz.cuda(), drop_mask=mask.cuda())
        tap = [{k: (v.cpu().clone() if torch.is_tensor(v) else v) for k, v in t.items()} for t in ops.sa_tap]
    finally:
        ops.sa_tap = None
    loss = spt.criterion(out, fwd.cuda()).mean()
    loss.backward()
    P = oracle.cast_params(state, torch.float64)
    diag = {}
    o64 = oracle.simple_pointnet_forward(xyz, P, mask.float(), True, None, argmax=tap[0]["argmax"], diag=diag)
    l64 = oracle.mse_rows(o64, fwd.double()).mean()
    l64.backward()
    assert max(diag["route_gap"]) <= 2e-6, diag
    assert abs(float(loss) - float(l64)) <= 1e-5
    assert float((out.detach().cpu().double() - o64.detach()).abs().max()) <= 5e-5
    num = den = 0.0
    for n, p in model.named_parameters():
        if n in ZERO_GRAD:
            continue
        d = p.grad.detach().cpu().double() - P[n].grad.reshape(p.shape)
        num, den = num + float((d * d).sum()), den + float((P[n].grad ** 2).sum())
    rel = math.sqrt(num / den)
    print(f"\n[simple B={B} N={N}] loss hip {float(loss):.8f} fp64 {float(l64):.8f} routed flat-grad relL2 {rel:.2e} "
          f"route gap {max(diag['route_gap']):.1e}")
    assert rel <= 5e-5        # measured 2.7e-6 (10,000 points x 16) / 8.1e-6 (777 x 3); round 3 gated this at 3e-3


def test_simple_pointnet_script_trains(tmp_path, monkeypatch, capsys):
    """main() on synthetic clouds at the configuration BASELINE.json quotes (256 points, batch 4): the loss falls, the
    reference's report lines are printed and the curve is written."""
    import simple_pointnet_train as spt
    monkeypatch.setattr(spt, "RES", tmp_path)
    tr, va, test_loss = spt.main(["--synthetic", "64", "--points", "256", "--batch", "4", "--epochs", "6"])
    text = capsys.readouterr().out
    assert len(tr) == 6 and len(va) == 6 and all(math.isfinite(v) for v in tr + va) and math.isfinite(test_loss)
    assert tr[-1] < tr[0]
    assert "Epoch [6/6] Train Loss:" in text and "Manual MSE loss for this sample:" in text and "Test Loss:" in text
    assert (tmp_path / "chair_simplepointnet_training_validation_loss.png").exists()
