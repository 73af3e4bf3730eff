"""GPU parity, set abstraction: PointNetSetAbstraction on the HIP kernels vs the CPU oracle evaluated
in float64 on the same float32 inputs (gate G2 of SURVEY 8d: <= 1e-5 relative to the tensor's max-abs
per operator; a whole SA block is three conv/BN/ReLU operators deep, so its bound is 3e-5) and vs the
vectors captured from the fp32 reference."""
import numpy as np
import pytest
import torch

from conftest import FLIP_MARGIN, ROUTE_GAP, routed_level, tap_to_routing

pytestmark = pytest.mark.gpu

SA_CFG = {
    "a": dict(npoint=32, nsample=16, in_channel=0, mlp=[32, 32, 64], group_all=False),
    "b": dict(npoint=16, nsample=16, in_channel=64, mlp=[32, 64, 64], group_all=False),
    "c": dict(npoint=None, nsample=None, in_channel=64, mlp=[64, 96, 128], group_all=True),
}


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _rel(got, ref):
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.abs(np.asarray(got, dtype=np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30))


def _module_from_fixture(g, tag):
    from models.pointnet_pp_8dir import PointNetSetAbstraction
    cfg = SA_CFG[tag]
    sa = PointNetSetAbstraction(cfg["npoint"], cfg["nsample"], cfg["in_channel"], cfg["mlp"], cfg["group_all"])
    sd = {k[len(tag) + 3:]: _t(g[k]) for k in g.files if k.startswith(f"{tag}_p.")}
    sa.load_state_dict(sd)
    return sa.cuda().train()


def _oracle_params(g, tag, dtype):
    P = {}
    for k in g.files:
        if k.startswith(f"{tag}_p."):
            v = _t(g[k])
            if v.is_floating_point():
                v = v.to(dtype)
                if "running" not in k:
                    v.requires_grad_(True)
            P["sa." + k[len(tag) + 3:]] = v
    return P


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_sa_forward_backward(oracle, golden, tag):
    g = golden("sa_small.npz")
    cfg = SA_CFG[tag]
    sa = _module_from_fixture(g, tag)
    xyz = _t(g[f"{tag}_xyz"])
    pts = _t(g[f"{tag}_pts"]) if f"{tag}_pts" in g.files else None
    centres = _t(g[f"{tag}_centres"].astype(np.int64)) if not cfg["group_all"] else None
    gy = _t(g[f"{tag}_gy"])

    pts_gpu = pts.cuda().requires_grad_(True) if pts is not None else None
    new_xyz, y = sa(xyz.cuda(), pts_gpu, centres.cuda() if centres is not None else None)
    y.backward(gy.cuda())

    # fp64 oracle on the same fp32 inputs
    P = _oracle_params(g, tag, torch.float64)
    pts64 = pts.double().requires_grad_(True) if pts is not None else None
    st = oracle.BNState()
    new_ref, y_ref, idx_ref = oracle.sa_forward(xyz, pts64, P, "sa", centres, cfg["nsample"], cfg["group_all"], True, st)
    (y_ref * gy.double()).sum().backward()

    assert torch.equal(new_xyz.cpu(), new_ref)
    assert _rel(y.detach().cpu(), y_ref.detach()) < 3e-5
    assert _rel(y.detach().cpu(), g[f"{tag}_y"]) < 1e-4                 # and the fp32 reference capture
    for name, p in sa.named_parameters():
        ref = P["sa." + name].grad.reshape(p.shape)
        if name.startswith("convs") and name.endswith("bias"):
            assert float(p.grad.abs().max()) == 0.0                       # exactly zero by construction
            assert float(ref.abs().max()) < 1e-9                          # and analytically zero in fp64
            continue
        assert _rel(p.grad.cpu(), ref) < 3e-5, name
        assert _rel(p.grad.cpu(), g[f"{tag}_g.{name}"]) < 2e-4, name      # fp32 reference capture (noisier)
    if pts is not None:
        assert _rel(pts_gpu.grad.cpu(), pts64.grad) < 3e-5
    for name, (rm, rv) in st.updates.items():
        key = name[3:]
        mod = sa.bns[int(key.split(".")[1])]
        assert _rel(mod.running_mean.cpu(), rm) < 1e-5
        assert _rel(mod.running_var.cpu(), rv) < 1e-5
        assert int(mod.num_batches_tracked) == 1


def test_sa_neighbour_sets_and_determinism(oracle, golden):
    from pnpp_hip import ops
    g = golden("sa_small.npz")
    sa = _module_from_fixture(g, "a")
    xyz = _t(g["a_xyz"]).cuda()
    centres = _t(g["a_centres"].astype(np.int64)).cuda()
    outs = []
    for _ in range(2):
        new_xyz, y, nbr = ops.set_abstraction(xyz, None, centres, 16, False, True, sa.convs, sa.bns, return_neighbours=True)
        y.sum().backward()
        outs.append((y.detach().clone(), [p.grad.clone() for p in sa.parameters()]))
        sa.zero_grad()
    want = oracle.knn_indices(oracle.index_points(xyz.cpu(), centres.cpu()), xyz.cpu(), 16)
    assert torch.equal(nbr.cpu().long(), want)
    assert torch.equal(outs[0][0], outs[1][0])                            # bitwise reproducible, forward ...
    assert all(torch.equal(a, b) for a, b in zip(outs[0][1], outs[1][1]))  # ... and backward (no atomics)


def test_sa_eval_mode_uses_running_stats(oracle, golden):
    g = golden("sa_small.npz")
    sa = _module_from_fixture(g, "b")
    with torch.no_grad():
        for bn in sa.bns:
            bn.running_mean.uniform_(-0.2, 0.2)
            bn.running_var.uniform_(0.5, 1.5)
    sa.eval()
    xyz, pts = _t(g["b_xyz"]), _t(g["b_pts"])
    centres = _t(g["b_centres"].astype(np.int64))
    P = {"sa." + k: v.detach().cpu().double() for k, v in sa.state_dict().items() if v.is_floating_point()}
    rm_before = sa.bns[0].running_mean.clone()
    with torch.no_grad():
        _, y = sa(xyz.cuda(), pts.cuda(), centres.cuda())
    _, y_ref, _ = oracle.sa_forward(xyz, pts.double(), P, "sa", centres, 16, False, training=False)
    assert _rel(y.cpu(), y_ref) < 3e-5
    assert torch.equal(rm_before, sa.bns[0].running_mean) and int(sa.bns[0].num_batches_tracked) == 0


def test_sa_ball_grouping_and_fps_sampler(oracle, golden):
    """The Demo's sampler / grouper (PointNet++Demo.py:8-70) plugged into the same block."""
    from models.pointnet_pp_8dir import PointNetSetAbstraction
    g = golden("sa_small.npz")
    torch.manual_seed(0)
    sa = PointNetSetAbstraction(24, 16, 0, [32, 32, 64]).cuda().train()
    sa.sampler, sa.grouper = "fps", ("ball", 0.35)
    xyz = _t(g["a_xyz"])
    torch.manual_seed(11)
    new_xyz, y = sa(xyz.cuda(), None)
    torch.manual_seed(11)
    start = torch.randint(0, xyz.shape[1], (xyz.shape[0],))
    centres = oracle.farthest_point_sample(xyz, 24, start.numpy())
    assert torch.equal(new_xyz.cpu(), oracle.index_points(xyz, centres))
    nbr = oracle.ball_query(0.35, 16, xyz, new_xyz.cpu())
    P = {"sa." + k: v.detach().cpu().double() for k, v in sa.state_dict().items() if v.is_floating_point()}
    _, y_ref, _ = oracle.sa_forward(xyz, None, P, "sa", centres, 16, False, True, None, neighbour_idx=nbr)
    assert _rel(y.detach().cpu(), y_ref) < 3e-5


def test_sa_config2_shapes(oracle):
    """The real SA1 / SA2 / SA3 shapes of config 2 at B=4 against the fp64 oracle."""
    from models.pointnet_pp_8dir import PointNetSetAbstraction
    torch.manual_seed(42)
    sa1 = PointNetSetAbstraction(128, 32, 0, [64, 64, 128]).cuda().train()
    sa2 = PointNetSetAbstraction(32, 32, 128, [128, 128, 256]).cuda().train()
    sa3 = PointNetSetAbstraction(None, None, 256, [256, 512, 1024], group_all=True).cuda().train()
    xyz, _, _, _ = oracle.synthetic_clouds(4, 1024, seed=5)
    torch.manual_seed(1)
    c1, c2 = oracle.replay_centres(4)
    from pnpp_hip import ops
    ops.sa_tap = []
    try:
        l1_xyz, l1 = sa1(xyz.cuda(), None, c1.cuda())
        l2_xyz, l2 = sa2(l1_xyz, l1, c2.cuda())
        _, l3 = sa3(l2_xyz, l2)
        routing = tap_to_routing(ops.sa_tap)
    finally:
        ops.sa_tap = None
    gy = torch.randn(l3.shape, generator=torch.Generator().manual_seed(2))
    l3.backward(gy.cuda())

    def params():
        P = {}
        for pre, m in (("sa1", sa1), ("sa2", sa2), ("sa3", sa3)):
            for k, v in m.state_dict().items():
                if v.is_floating_point():
                    t = v.detach().cpu().double()
                    P[f"{pre}.{k}"] = t.requires_grad_(True) if "running" not in k else t
        return P

    def errors(P):
        worst, num, den = 0.0, 0.0, 0.0
        for pre, m in (("sa1", sa1), ("sa2", sa2), ("sa3", sa3)):
            for k, p in m.named_parameters():
                ref = P[f"{pre}.{k}"].grad.reshape(p.shape)
                if k.startswith("convs") and k.endswith("bias"):
                    continue
                if float(ref.abs().max()) < 1e-9:
                    # structurally zero: e.g. sa{1,2}.bns.2.bias when every pooled maximum is positive,
                    # the shift is then removed by the next layer's BatchNorm; fp32 can only produce noise here
                    assert float(p.grad.abs().max()) < 1e-3, (pre, k)
                    continue
                num += float((p.grad.cpu().double() - ref).pow(2).sum())
                den += float(ref.pow(2).sum())
                worst = max(worst, _rel(p.grad.cpu(), ref))
        return np.sqrt(num / den), worst

    # (1) float64 handed every discrete decision of the HIP path (neighbour order, max-pool routing, ReLU decisions of all nine layers):
    # a smooth function of rounding -- nine BatchNorms deep, tensor by tensor
    P, diag = params(), {}
    feat = oracle.backbone_forward(xyz, P, [c1, c2], True, None, routing=routing, diag=diag)
    (feat * gy.double().reshape(feat.shape)).sum().backward()
    assert max(diag["route_gap"]) <= ROUTE_GAP and max(diag["relu_flip_margin"]) <= FLIP_MARGIN, diag
    assert _rel(l3.detach().cpu().reshape(feat.shape), feat.detach()) < 1e-5
    l2r, worst_r = errors(P)
    # (2) float64 on its own decisions: what the discrete decisions are worth (diagnostic; the gate only bounds O(1) errors)
    P = params()
    feat = oracle.backbone_forward(xyz, P, [c1, c2], True, None)
    (feat * gy.double().reshape(feat.shape)).sum().backward()
    l2o, worst_o = errors(P)
    print(f"\nbackbone B=4: flat gradient rel L2 {l2r:.2e} / worst per-tensor rel-to-max {worst_r:.2e} with the HIP path's decisions injected "
          f"({sum(diag['relu_flips'])} ReLU decisions differ from float64's own); {l2o:.2e} / {worst_o:.2e} on float64's own decisions")
    assert l2r < 2e-5 and worst_r < 2e-5, (l2r, worst_r)   # measured: see DESIGN section 5
    assert l2o < 1e-2, l2o                                  # every differing decision is an O(1) change of one element's gradient


def test_sa1_shape_backward_in_eval_mode(oracle):
    """Backward through the SA1 shape with BatchNorm in eval mode: the BatchNorm-backward transform degenerates to dZ = g dY (its
    z-coefficient is exactly zero), which the fused backward kernel of the last layer cannot fold into its weight panel -- the
    launch takes its other form (gemm_wsp_kernels.hip).  Against the float64 oracle with the kernels' max-pool routing and ReLU decisions."""
    from models.pointnet_pp_8dir import PointNetSetAbstraction
    torch.manual_seed(7)
    sa = PointNetSetAbstraction(128, 32, 0, [64, 64, 128]).cuda()
    with torch.no_grad():
        for bn in sa.bns:
            bn.running_mean.uniform_(-0.2, 0.2)
            bn.running_var.uniform_(0.5, 1.5)
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.2, 0.2)
    sa.eval()
    xyz, _, _, _ = oracle.synthetic_clouds(4, 1024, seed=9)
    torch.manual_seed(3)
    c1, _ = oracle.replay_centres(4)
    gy = torch.randn(4, 128, 128, generator=torch.Generator().manual_seed(4))
    res, diag = routed_level(oracle, sa, xyz, None, c1, gy, 32, False, training=False)
    print("\n[sa1 eval] " + ", ".join(f"{k} {v:.1e}" for k, v in res.items()))
    assert max(res.values()) <= 1e-5, res


@pytest.mark.parametrize("training,B", [(True, 32), (False, 32), (True, 36)])
def test_sa2_shape_backward_of_the_256_channel_last_layer(oracle, training, B):
    """The SA2 shape (128 features, 32 centres x 32 neighbours, [128, 128, 256]) at batch 32 = 32,768 rows: the last layer's fused backward
    product runs on gemm_wsq_kernel (BatchNorm-backward folded into the weight panel; in eval mode its z-coefficient is exactly zero
    and the launch takes the k-form).  Every parameter gradient and the feature gradient against the float64 oracle."""
    from models.pointnet_pp_8dir import PointNetSetAbstraction
    torch.manual_seed(13)
    sa = PointNetSetAbstraction(32, 32, 128, [128, 128, 256]).cuda()
    with torch.no_grad():
        for bn in sa.bns:
            bn.running_mean.uniform_(-0.2, 0.2)
            bn.running_var.uniform_(0.5, 1.5)
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.2, 0.2)
    sa.train(training)
    N = 128   # (B = 36: 576 tiles over 128 workers -- some take four, some five)
    g = torch.Generator().manual_seed(8)
    xyz = torch.rand(B, N, 3, generator=g) * 2 - 1
    feats = torch.randn(B, N, 128, generator=g)
    c = torch.stack([torch.randperm(N, generator=g)[:32] for _ in range(B)])
    gy = torch.randn(B, 32, 256, generator=torch.Generator().manual_seed(9))
    # every parameter gradient, the feature gradient and the output, tensor by tensor, with the HIP path's decisions injected
    # (round 3 gated these at 1e-2 relative L2 "for O(1) errors": a float32 ReLU decision or max-pool tie falls the other way than
    # float64's about once per layer and pass, each worth 5e-4 ... 3e-3 of a tensor's norm)
    res, diag = routed_level(oracle, sa, xyz, feats, c, gy, 32, False, training=training)
    print(f"\n[sa2 shape training={training} B={B}] flips {diag['relu_flips']} " + ", ".join(f"{k} {v:.1e}" for k, v in res.items()))
    assert max(res.values()) <= 1e-5, res


def test_sa2_shape_second_form_of_the_kernel_in_a_child_process():
    """gemm_wsq2_kernel (PNPP_WSQ_FORM=2: weight panel in registers, two tile images filled by LDS-DMA; measured equal, not the default)
    stays under the same gate.  The switch is read once per process, so the test above runs again in a fresh child."""
    import os, subprocess, sys
    env = dict(os.environ, PNPP_WSQ_FORM="2")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-p", "no:cacheprovider", "-k",
                        "test_sa2_shape_backward_of_the_256_channel_last_layer"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "3 passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("B,N", [(4, 1024), (9, 640)])
def test_sa1_backward_of_layers_0_and_1_from_the_coordinates(oracle, B, N):
    """Train-mode backward through the SA1 shape (D = 0, 32 neighbours, [64, 64, 128]) at 8192+ rows: layer 1's backward rebuilds
    Z_0 from the relative coordinates, never writes dY_0, and layer 0's parameter gradients come from per-channel sums and the
    coordinate moments (gemm_wsx_kernels.hip).  Every parameter gradient against the float64 oracle, tensor by tensor."""
    from models.pointnet_pp_8dir import PointNetSetAbstraction
    torch.manual_seed(11)
    sa = PointNetSetAbstraction(128, 32, 0, [64, 64, 128]).cuda().train()
    with torch.no_grad():
        for bn in sa.bns:
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
    xyz, _, _, _ = oracle.synthetic_clouds(B, N, seed=21)
    g = torch.Generator().manual_seed(5)
    c1 = torch.stack([torch.randperm(N, generator=g)[:128] for _ in range(B)])
    gy = torch.randn(B, 128, 128, generator=torch.Generator().manual_seed(6))
    res, diag = routed_level(oracle, sa, xyz, None, c1, gy, 32, False, training=True)
    print(f"\n[sa1 coordinates B={B} N={N}] flips {diag['relu_flips']} " + ", ".join(f"{k} {v:.1e}" for k, v in res.items()))
    assert max(res.values()) <= 1e-5, res


@pytest.mark.parametrize("c0,npoint,nsample,n", [(256, 96, 64, 256), (512, 40, 16, 128), (64, 160, 32, 512)])
def test_sa_layer0_convolved_before_the_gather(oracle, c0, npoint, nsample, n):
    """Grouped levels with input features run layer 0 on the source points and gather afterwards (P[idx] + W_xyz (x - c),
    csrc/gemm_kernels.hip gather_rel_stats / scatter_dz): wide layers (2 and 4 channel chunks per lane in the scatter),
    both the on-the-fly dZ (B*S*K > 4096 rows) and the materialised one, against the fp64 oracle of the reference's
    gather-then-convolve order (pointnet_pp_8dir.py:28-43)."""
    from models.pointnet_pp_8dir import PointNetSetAbstraction
    torch.manual_seed(c0)
    B, D = 3, 32
    sa = PointNetSetAbstraction(npoint, nsample, D, [c0, 64, 96]).cuda().train()
    g = torch.Generator().manual_seed(7)
    xyz = torch.rand(B, n, 3, generator=g) * 2 - 1
    pts = torch.randn(B, n, D, generator=g)
    centres = torch.stack([torch.randperm(n, generator=g)[:npoint] for _ in range(B)])
    pts_gpu = pts.cuda().requires_grad_(True)
    _, y = sa(xyz.cuda(), pts_gpu, centres.cuda())
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy.cuda())

    P = {}
    for k, v in sa.state_dict().items():
        if v.is_floating_point():
            t = v.detach().cpu().double()
            P["sa." + k] = t.requires_grad_(True) if "running" not in k else t
    pts64 = pts.double().requires_grad_(True)
    _, y_ref, _ = oracle.sa_forward(xyz, pts64, P, "sa", centres, nsample, False, True, oracle.BNState())
    (y_ref * gy.double()).sum().backward()
    assert _rel(y.detach().cpu(), y_ref.detach()) < 3e-5
    assert _rel(pts_gpu.grad.cpu(), pts64.grad) < 5e-5
    for name, p in sa.named_parameters():
        if name.startswith("convs") and name.endswith("bias"):
            continue
        assert _rel(p.grad.cpu(), P["sa." + name].grad.reshape(p.shape)) < 5e-5, name


@pytest.mark.parametrize("B,N,S1,K1,S2,K2", [(32, 1024, 128, 32, 32, 32), (3, 777, 96, 24, 40, 16), (2, 2048, 128, 32, 32, 32)])
def test_two_levels_grouped_in_one_launch_equal_the_per_level_path(B, N, S1, K1, S2, K2):
    """ops.group_pair (pnpp_sa_group_pair): the centre gathers and neighbour searches of sa1 and sa2 as ONE launch ahead of both
    MLPs -- level 2 searches among level 1's centres, rows of the same cloud -- must give, bit for bit, what the two
    per-level searches give (models/pointnet_pp_8dir.py:28-31 twice): neighbour indices, centre coordinates, both outputs and
    the gradients that flow back through them."""
    from models.pointnet_pp_8dir import PointNetSetAbstraction, stacked_levels
    from pnpp_hip import ops
    torch.manual_seed(3)
    sa1 = PointNetSetAbstraction(S1, K1, 0, [32, 32, 64]).cuda().train()
    sa2 = PointNetSetAbstraction(S2, K2, 64, [64, 64, 128]).cuda().train()
    g = torch.Generator().manual_seed(N)
    xyz = (torch.rand(B, N, 3, generator=g) * 2 - 1).cuda()
    c1 = torch.stack([torch.randperm(N, generator=g)[:S1] for _ in range(B)]).cuda()
    c2 = torch.stack([torch.randperm(S1, generator=g)[:S2] for _ in range(B)]).cuda()

    def run(paired):
        for m in (sa1, sa2):
            m.zero_grad()
        ops.sa_tap = []
        try:
            if paired:
                l1_xyz, l1, l2_xyz, l2 = stacked_levels(sa1, sa2, xyz, c1, c2)
            else:
                l1_xyz, l1 = sa1(xyz, None, c1)
                l2_xyz, l2 = sa2(l1_xyz, l1, c2)
            taps = [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in t.items()} for t in ops.sa_tap]
        finally:
            ops.sa_tap = None
        (l2 * torch.linspace(-1, 1, l2.numel(), device=l2.device).view_as(l2)).sum().backward()
        grads = [p.grad.clone() for m in (sa1, sa2) for p in m.parameters()]
        return l1_xyz, l1, l2_xyz, l2, taps, grads

    sd = [{k: v.clone() for k, v in m.state_dict().items()} for m in (sa1, sa2)]
    a = run(False)
    for m, s in zip((sa1, sa2), sd):
        m.load_state_dict(s)
    b = run(True)
    for i in range(4):
        assert torch.equal(a[i], b[i]), i
    for ta, tb in zip(a[4], b[4]):
        assert torch.equal(ta["neighbours"], tb["neighbours"]) and torch.equal(ta["argmax"], tb["argmax"])
    for ga, gb in zip(a[5], b[5]):
        assert torch.equal(ga, gb)
