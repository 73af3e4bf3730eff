#!/usr/bin/env python3
"""oracle/make_golden.py -- TEST INFRASTRUCTURE ONLY.

Captures golden input/output vectors from the *reference itself* (imported read-only from
/root/reference in the build container) and writes them to tests/golden/ as small .npz
fixtures.  The reference never travels to the GPU box; these fixtures and oracle/ do.

  python oracle/make_golden.py            # regenerate every fixture

How the reference is driven (no reference source is copied):
  * models.*            imported as-is (sys.path insert).
  * PointNet++Demo.py   loaded with importlib (its file name is not a legal module name).
  * loss functions      live inside training scripts that start training on import
                        (train_single_peak_vonMises_KL.py:39 iterates the dataset at module
                        level), so the FunctionDef nodes `kl_von_mises` / `match_loss` /
                        `kl_loss_per_sample_from_logits` are parsed out with `ast` at run
                        time and compiled on their own -- the reference's own function
                        objects are what produce the vectors.
  * randperm centre draws (pointnet_pp_8dir.py:28) are replayed by seeding the CPU generator
    and repeating the same draws; dropout is removed (nn.Identity) or replaced by a module
    that applies a stored mask.
  * The fp64 captures run the same reference modules after .double(); only the neighbour
    search is wrapped so that its inputs are cast to float32 first (the index sets must be
    those of the fp32 recipe).
"""
import ast
import importlib.util
import json
import math
import os
import re
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from oracle import restatement as R  # noqa: E402  (synthetic inputs only)


def ref_functions(script, names, extra_ns=None):
    """Compile selected top-level functions of a reference script without running the script."""
    src = open(os.path.join(REF, script), encoding="utf-8").read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(keep) == len(names), (script, names)
    mod = ast.Module(body=keep, type_ignores=[])
    from scipy.optimize import linear_sum_assignment
    import torch.nn.functional as F
    ns = {"torch": torch, "math": math, "np": np, "F": F, "device": torch.device("cpu"),
          "linear_sum_assignment": linear_sum_assignment}
    ns.update(extra_ns or {})
    exec(compile(mod, os.path.join(REF, script), "exec"), ns)
    return [ns[n] for n in names]


def load_demo():
    spec = importlib.util.spec_from_file_location("pnpp_demo", os.path.join(REF, "PointNet++Demo.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrs)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


def n(t):
    return t.detach().cpu().numpy().copy()


# ------------------------------------------------------------------------------------------
def golden_index():
    from models import base
    demo = load_demo()
    g = torch.Generator().manual_seed(20251114)
    out = {}
    # (1) square_distance, three shapes
    for i, (B, S, N) in enumerate([(2, 8, 64), (1, 16, 333), (2, 32, 128)]):
        src = torch.rand(B, S, 3, generator=g) * 2 - 1
        dst = torch.rand(B, N, 3, generator=g) * 2 - 1
        out[f"sq{i}_src"], out[f"sq{i}_dst"] = n(src), n(dst)
        out[f"sq{i}_out"] = n(base.square_distance(src, dst))
    # (2) kNN sets at the SA1 / SA2 shapes of config 2, centres injected
    xyz, _, _, _ = R.synthetic_clouds(2, 1024, seed=77)
    c1 = torch.stack([torch.randperm(1024, generator=g)[:128] for _ in range(2)])
    new1 = base.index_points(xyz, c1)
    idx1 = base.query_ball_point(new1, xyz, 32)
    c2 = torch.stack([torch.randperm(128, generator=g)[:32] for _ in range(2)])
    new2 = base.index_points(new1, c2)
    idx2 = base.query_ball_point(new2, new1, 32)
    out.update(knn_xyz=n(xyz), knn_c1=n(c1).astype(np.int16), knn_c2=n(c2).astype(np.int16),
               knn_idx1_sorted=np.sort(n(idx1), -1).astype(np.int16),
               knn_idx2_sorted=np.sort(n(idx2), -1).astype(np.int16))
    # a ragged shape (N not a multiple of 64, k not 32)
    xr = torch.rand(3, 200, 3, generator=g)
    cr = torch.stack([torch.randperm(200, generator=g)[:10] for _ in range(3)])
    out.update(knnr_xyz=n(xr), knnr_c=n(cr).astype(np.int16),
               knnr_idx_sorted=np.sort(n(base.query_ball_point(base.index_points(xr, cr), xr, 7)), -1).astype(np.int16))
    # (3) Demo FPS with injected start indices: replay the randint draw
    torch.manual_seed(5)
    fps = demo.farthest_point_sample(xyz, 128)
    torch.manual_seed(5)
    start = torch.randint(0, 1024, (2,), dtype=torch.long)
    assert torch.equal(fps[:, 0], start)
    out.update(fps_start=n(start).astype(np.int32), fps_idx=n(fps).astype(np.int16))
    torch.manual_seed(6)
    fr = demo.farthest_point_sample(xr, 17)
    out.update(fpsr_start=n(fr[:, 0]).astype(np.int32), fpsr_idx=n(fr).astype(np.int16))
    # (4) Demo radius query
    newf = demo.index_points(xyz, fps)
    for r in (0.2, 0.4):
        out[f"ball_{r}"] = n(demo.query_ball_point(r, 32, xyz, newf)).astype(np.int16)
    out["ballr_0.3"] = n(demo.query_ball_point(0.3, 5, xr, demo.index_points(xr, fr))).astype(np.int16)
    save("index.npz", **out)


# ------------------------------------------------------------------------------------------
def _state(mod):
    return {k: n(v) for k, v in mod.state_dict().items()}


def golden_sa():
    """PointNetSetAbstraction forward + backward on three small configs (fp32, reference)."""
    from models.pointnet_pp_8dir import PointNetSetAbstraction
    out = {}
    g = torch.Generator().manual_seed(99)
    xyz, _, _, _ = R.synthetic_clouds(4, 256, seed=11)
    cfgs = {
        "a": dict(npoint=32, nsample=16, in_channel=0, mlp=[32, 32, 64]),
        "b": dict(npoint=16, nsample=16, in_channel=64, mlp=[32, 64, 64]),
        "c": dict(npoint=None, nsample=None, in_channel=64, mlp=[64, 96, 128], group_all=True),
    }
    feats = None
    cur_xyz = xyz
    for tag, cfg in cfgs.items():
        torch.manual_seed(100 + ord(tag))
        sa = PointNetSetAbstraction(cfg["npoint"], cfg["nsample"], cfg["in_channel"], cfg["mlp"],
                                    cfg.get("group_all", False))
        # non-trivial BN affine parameters so d(gamma), d(beta) are exercised
        with torch.no_grad():
            for bn in sa.bns:
                bn.weight.copy_(torch.rand(bn.weight.shape, generator=g) + 0.5)
                bn.bias.copy_(torch.rand(bn.bias.shape, generator=g) - 0.5)
        sa.train()
        pts = None
        if cfg["in_channel"]:
            pts = torch.randn(cur_xyz.shape[0], cur_xyz.shape[1], cfg["in_channel"], generator=g).requires_grad_(True)
        st0 = _state(sa)
        torch.manual_seed(1000 + ord(tag))
        new_xyz, y = sa(cur_xyz, pts)
        if not cfg.get("group_all", False):
            torch.manual_seed(1000 + ord(tag))
            centres = torch.stack([torch.randperm(cur_xyz.shape[1])[:cfg["npoint"]] for _ in range(cur_xyz.shape[0])])
            assert torch.equal(new_xyz, torch.stack([cur_xyz[b, centres[b]] for b in range(cur_xyz.shape[0])]))
            out[f"{tag}_centres"] = n(centres).astype(np.int16)
        gy = torch.randn(y.shape, generator=g)
        (y * gy).sum().backward()
        out[f"{tag}_xyz"] = n(cur_xyz)
        if pts is not None:
            out[f"{tag}_pts"], out[f"{tag}_dpts"] = n(pts), n(pts.grad)
        out[f"{tag}_y"], out[f"{tag}_gy"] = n(y), n(gy)
        for k, v in st0.items():
            out[f"{tag}_p.{k}"] = v
        for k, p in sa.named_parameters():
            out[f"{tag}_g.{k}"] = n(p.grad)
        for k, v in _state(sa).items():
            if "running" in k or "num_batches" in k:
                out[f"{tag}_after.{k}"] = v
        cur_xyz = new_xyz.detach()
    save("sa_small.npz", **out)


# ------------------------------------------------------------------------------------------
class _Mask(nn.Module):
    def __init__(self, mask, p):
        super().__init__()
        self.mask, self.p = mask, p

    def forward(self, x):
        return x * self.mask.to(x.dtype) / (1.0 - self.p) if self.training else x


def _grad_summary(model, pos_gen_seed=3):
    """Per-parameter L2 norm, sum and 16 sampled entries at fixed pseudo-random positions."""
    g = torch.Generator().manual_seed(pos_gen_seed)
    out = {}
    for k, p in model.named_parameters():
        gr = p.grad.detach().double().flatten()
        pos = torch.randint(0, gr.numel(), (16,), generator=g)
        out[f"gn.{k}"] = np.array([gr.norm().item(), gr.sum().item()])
        out[f"gp.{k}"] = n(pos).astype(np.int64)
        out[f"gs.{k}"] = n(gr[pos])
    return out


def _param_checksums(model):
    return {f"ck.{k}": np.array([v.double().sum().item(), v.double().abs().sum().item()])
            for k, v in model.state_dict().items() if v.is_floating_point()}


def golden_e2e():
    import models.pointnet_pp_8dir as m8
    from models.pointnet_pp_vonMises import PointNetPPVonMises
    from models.pointnet_pp_mvM import PointNetPPMvM
    from models.pointnet_pp_8dir import PointNetPP8Dir, DIRS_8
    (kl_single,) = ref_functions("train_single_peak_vonMises_KL.py", ["kl_von_mises"])
    kl_multi, match_loss = ref_functions("train_multi_peaks_vonMises_KL.py", ["kl_von_mises", "match_loss"])
    (soft_ce,) = ref_functions("train_8dir_KL.py", ["kl_loss_per_sample_from_logits"])
    orig_q = m8.query_ball_point

    def q32(new_xyz, xyz, k):
        return orig_q(new_xyz.float(), xyz.float(), k)

    B, N = 8, 1024
    xyz, mu_gt, kappa_gt, fwd = R.synthetic_clouds(B, N, seed=1234)
    out = dict(xyz_ck=np.array([xyz.double().sum().item(), xyz.double().abs().sum().item()]),
               mu_gt=n(mu_gt), kappa_gt=n(kappa_gt))
    gmask = torch.Generator().manual_seed(8)
    mask = (torch.rand(B, 256, generator=gmask) < 0.5).float()
    out["drop_mask"] = n(mask).astype(np.uint8)

    for dt_name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        m8.query_ball_point = q32 if dt == torch.float64 else orig_q
        for variant in ("nodrop", "mask"):
            torch.manual_seed(42)
            model = PointNetPPVonMises().to(dt)
            if dt == torch.float32 and variant == "nodrop":
                out.update(_param_checksums(model))
            model.drop = nn.Identity() if variant == "nodrop" else _Mask(mask, 0.5)
            model.train()
            torch.manual_seed(4242)
            mu, kappa = model(xyz.to(dt))
            loss_vec = kl_single(mu, kappa, mu_gt.to(dt), kappa_gt.to(dt))
            loss = loss_vec.mean()
            loss.backward()
            tag = f"vm_{dt_name}_{variant}"
            out[f"{tag}.mu"], out[f"{tag}.kappa"] = n(mu), n(kappa)
            out[f"{tag}.loss_vec"], out[f"{tag}.loss"] = n(loss_vec), np.array(loss.item())
            for k, v in _grad_summary(model).items():
                out[f"{tag}.{k}"] = v
            if variant == "nodrop":
                for k, v in model.state_dict().items():
                    if "running" in k:
                        out[f"{tag}.after.{k}"] = n(v)
    torch.manual_seed(4242)
    cs = R.replay_centres(B)
    out["centres1"], out["centres2"] = n(cs[0]).astype(np.int16), n(cs[1]).astype(np.int16)

    # ---- multi-peak model + match_loss (config 3), heads perturbed away from their zero init
    K_gt = torch.tensor([1, 2, 4, 4, 0, 2, 1, 4])
    side = torch.stack([-fwd[:, 2], torch.zeros(B), fwd[:, 0]], 1)
    peaks = torch.stack([fwd, -fwd, side, -side], 1)                       # (B,4,3)
    vm_gt = torch.zeros(B, 4, 3)
    for b in range(B):
        k = int(K_gt[b])
        for j in range(k):
            vm_gt[b, j] = torch.tensor([math.atan2(peaks[b, j, 0], -peaks[b, j, 2]), 8.0, 1.0 / k])
    out["mvm_vm_gt"], out["mvm_K"] = n(vm_gt), n(K_gt)
    for dt_name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        m8.query_ball_point = q32 if dt == torch.float64 else orig_q
        torch.manual_seed(42)
        model = PointNetPPMvM()
        torch.manual_seed(7)
        with torch.no_grad():
            model.head_pi.weight.normal_(0, 0.05)
            model.head_mu.weight.normal_(0, 0.05)
            model.head_mu.bias.normal_(0, 0.05)
        if dt_name == "f32":
            out.update({f"mvm_{k}": v for k, v in _param_checksums(model).items()})
        model = model.to(dt)
        model.drop = nn.Identity()
        model.train()
        torch.manual_seed(4242)
        mu, kappa, w = model(xyz.to(dt))
        lv = match_loss(mu, kappa, w, vm_gt.to(dt), None, K_gt)
        # match_loss allocates its result as float32 (torch.zeros default dtype), keep as returned
        loss = lv.mean()
        loss.backward()
        tag = f"mvm_{dt_name}"
        out[f"{tag}.mu"], out[f"{tag}.kappa"], out[f"{tag}.w"] = n(mu), n(kappa), n(w)
        out[f"{tag}.loss_vec"], out[f"{tag}.loss"] = n(lv), np.array(loss.item())
        for k, v in _grad_summary(model).items():
            out[f"{tag}.{k}"] = v

    # ---- 8-direction model + soft-label CE (config 4 head; small batch here)
    prob8 = torch.relu(fwd @ DIRS_8.t())
    prob8 = prob8 / prob8.sum(1, keepdim=True)
    out["dir8_prob"] = n(prob8)
    for dt_name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        m8.query_ball_point = q32 if dt == torch.float64 else orig_q
        torch.manual_seed(42)
        model = PointNetPP8Dir().to(dt)
        model.drop = nn.Identity()
        model.train()
        torch.manual_seed(4242)
        logits = model(xyz.to(dt))
        lv = soft_ce(logits, prob8.to(dt))
        lv.mean().backward()
        tag = f"dir8_{dt_name}"
        out[f"{tag}.logits"], out[f"{tag}.loss_vec"] = n(logits), n(lv)
        for k, v in _grad_summary(model).items():
            out[f"{tag}.{k}"] = v
    m8.query_ball_point = orig_q
    save("e2e.npz", **out)


# ------------------------------------------------------------------------------------------
def golden_kl():
    (kl_single,) = ref_functions("train_single_peak_vonMises_KL.py", ["kl_von_mises"])
    kl_multi, match_loss = ref_functions("train_multi_peaks_vonMises_KL.py", ["kl_von_mises", "match_loss"])
    (soft_ce,) = ref_functions("train_8dir_KL.py", ["kl_loss_per_sample_from_logits"])
    out = {}
    kp = torch.tensor([0.0, 1e-7, 1e-6, 2e-6, 1e-3, 0.5, 1.2209635, 5.0, 20.0, 50.0, 80.0])
    kq = torch.tensor([0.0, 0.5, 8.0, 30.0, 80.0])
    mp = torch.tensor([-3.1, -1.0, 0.0, 0.7, 3.0])
    mq = torch.tensor([-2.5, 0.3, 3.1])
    grid = torch.cartesian_prod(mp, kp, mq, kq)
    g = torch.Generator().manual_seed(21)
    rnd = torch.stack([(torch.rand(400, generator=g) * 2 - 1) * math.pi, torch.rand(400, generator=g) * 80,
                       (torch.rand(400, generator=g) * 2 - 1) * math.pi, torch.rand(400, generator=g) * 80], 1)
    cases = torch.cat([grid, rnd], 0).float()
    out["single_in"] = n(cases)
    for dt_name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        c = cases.to(dt)
        a, b = c[:, 0].clone().requires_grad_(True), c[:, 1].clone().requires_grad_(True)
        v = kl_single(a, b, c[:, 2], c[:, 3])
        v.sum().backward()
        out[f"single_{dt_name}"] = np.stack([n(v), n(a.grad), n(b.grad)], 1)
        a, b = c[:, 0].clone().requires_grad_(True), c[:, 1].clone().requires_grad_(True)
        v = kl_multi(a, b, c[:, 2], c[:, 3])
        v.sum().backward()
        out[f"multi_{dt_name}"] = np.stack([n(v), n(a.grad), n(b.grad)], 1)
    # match_loss cases
    Bm = 96
    mu = (torch.rand(Bm, 4, generator=g) * 2 - 1) * math.pi
    kap = torch.rand(Bm, 4, generator=g) * 30 + 0.05
    w = torch.softmax(torch.randn(Bm, 4, generator=g), -1)
    vm = torch.zeros(Bm, 4, 3)
    vm[..., 0] = (torch.rand(Bm, 4, generator=g) * 2 - 1) * math.pi
    vm[..., 1] = torch.where(torch.rand(Bm, 4, generator=g) < 0.2, torch.zeros(Bm, 4), torch.full((Bm, 4), 8.0))
    K = torch.tensor([0, 1, 2, 3, 4, 4] * 16)
    out.update(match_mu=n(mu), match_kappa=n(kap), match_w=n(w), match_vm=n(vm), match_K=n(K))
    for dt_name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        a, b, c = (t.to(dt).clone().requires_grad_(True) for t in (mu, kap, w))
        lv = match_loss(a, b, c, vm.to(dt), None, K)
        lv.sum().backward()
        out[f"match_{dt_name}_loss"] = n(lv)
        out[f"match_{dt_name}_grads"] = np.stack([n(a.grad), n(b.grad), n(c.grad)], 0)
    # soft-label CE
    lg = torch.randn(32, 8, generator=g) * 3
    pt = torch.softmax(torch.randn(32, 8, generator=g), -1)
    pt[0] = 0.125
    lgr = lg.clone().requires_grad_(True)
    v = soft_ce(lgr, pt)
    v.sum().backward()
    out.update(ce_logits=n(lg), ce_p=n(pt), ce_loss=n(v), ce_grad=n(lgr.grad))
    save("kl.npz", **out)


# ------------------------------------------------------------------------------------------
def golden_debug_log(max_blocks=200):
    """Subsample of the reference's own run log (results/multi_peak_vonMises_KL_debug/debug_log.txt):
    inputs printed at train_multi_peaks_vonMises_KL_debug.py:89-95, matched cost at :111."""
    path = os.path.join(REF, "results/multi_peak_vonMises_KL_debug/debug_log.txt")
    txt = open(path, encoding="utf-8").read()
    blocks = re.split(r"(?m)^\[Batch \d+\] K = ", txt)[1:]

    def arr(s):
        return [float(t) for t in re.sub(r"[\[\],]", " ", s).split()]

    parsed = {1: [], 2: [], 4: []}
    for blk in blocks:
        K = int(blk.split("\n", 1)[0])
        flat = " ".join(blk.split("\n")[1:])
        m = re.search(r"μp:(.*?)κp:(.*?)wp:(.*?)μg:(.*?)κg:(.*?)matched cost:(.*?)matched_ws:(.*?)sum:", flat)
        if not m or K not in parsed:
            continue
        f = [arr(x) for x in m.groups()]
        if all(len(v) == K for v in f):
            parsed[K].append(f)
    print({k: len(v) for k, v in parsed.items()})
    out = {}
    quota = {1: max_blocks // 4, 2: max_blocks // 4, 4: max_blocks // 2}
    for K, lst in parsed.items():
        step = max(1, len(lst) // quota[K])
        pick = lst[::step][:quota[K]]
        a = np.array(pick, dtype=np.float64)           # (n, 7, K)
        out[f"K{K}"] = a
    save("debug_log_kat.npz", **out)


def golden_f3():
    """The other set-abstraction models: PointNetPP, PointNetPPFwd, PointNetPPXYZ,
    PointNetPPXYZ_Schedmit run by the reference in float32 and float64, with the loss expressions of their training
    scripts: nn.MSELoss on the raw vector (train_8dir.py:53,67), proj_probs + MSELoss (train_multi_8dir.py:41-44,80,100;
    proj_probs is the reference's own function object), and the inline axis-pair loss of train.py:183-187."""
    import importlib
    from models.pointnet_pp_8dir import DIRS_8
    proj = {dt: ref_functions("train_multi_8dir.py", ["proj_probs"], {"DIRS_8_T": DIRS_8.to(dt)})[0]
            for dt in (torch.float32, torch.float64)}   # the script's global DIRS_8_T, in the run's dtype
    crit = nn.MSELoss()
    B, N = 8, 1024
    xyz, _, _, fwd = R.synthetic_clouds(B, N, seed=1234)
    up = torch.tensor([0.0, 1.0, 0.0]).expand(B, 3).contiguous()
    side = torch.stack([-fwd[:, 2], torch.zeros(B), fwd[:, 0]], 1)
    prob8 = torch.relu(fwd @ DIRS_8.t())
    prob8 = prob8 / prob8.sum(1, keepdim=True)
    gmask = torch.Generator().manual_seed(8)
    mask = (torch.rand(B, 256, generator=gmask) < 0.5).float()
    out = dict(xyz_ck=np.array([xyz.double().sum().item(), xyz.double().abs().sum().item()]), fwd=n(fwd), up=n(up),
               side=n(side), prob8=n(prob8), drop_mask=n(mask).astype(np.uint8))
    specs = [("pp", "models.pointnet_pp", "PointNetPP"), ("fwd", "models.pointnet_pp_Fwd", "PointNetPPFwd"),
             ("xyz", "models.Pointnet_pp_xyz", "PointNetPPXYZ"), ("sch", "models.Pointnet_pp_xyz_Schedmit", "PointNetPPXYZ_Schedmit")]
    for tag0, modname, cls in specs:
        mod = importlib.import_module(modname)
        orig_q = mod.query_ball_point

        def q32(new_xyz, x, k, _o=orig_q):
            return _o(new_xyz.float(), x.float(), k)

        for dt_name, dt in (("f32", torch.float32), ("f64", torch.float64)):
            mod.query_ball_point = q32 if dt == torch.float64 else orig_q
            for variant in ("nodrop", "mask"):
                torch.manual_seed(42)
                model = getattr(mod, cls)()
                if dt_name == "f32" and variant == "nodrop":
                    out.update({f"{tag0}_{k}": v for k, v in _param_checksums(model).items()})
                model = model.to(dt)
                model.drop = nn.Identity() if variant == "nodrop" else _Mask(mask, 0.5)
                model.train()
                torch.manual_seed(4242)
                res = model(xyz.to(dt))
                tag = f"{tag0}_{dt_name}_{variant}"
                if tag0 == "pp":
                    loss = crit(res, fwd.to(dt))                                    # train_8dir.py:67
                    out[f"{tag}.out"] = n(res)
                elif tag0 == "fwd":
                    pred = proj[dt](res)                                            # train_multi_8dir.py:99
                    loss = crit(pred, prob8.to(dt))                                 # :100
                    out[f"{tag}.out"], out[f"{tag}.probs"] = n(res), n(pred)
                else:
                    va, vb = res
                    ga, gb = (side, up) if tag0 == "xyz" else (up, fwd)
                    pred_loss = (crit(va, ga.to(dt)) + crit(vb, gb.to(dt))) / 2.0   # train.py:183
                    dot_prod = (va * vb).sum(dim=1)                                 # :184
                    orth_loss = dot_prod.pow(2).mean()                              # :185
                    loss = pred_loss + 0.1 * orth_loss                              # :186-187
                    out[f"{tag}.out_a"], out[f"{tag}.out_b"] = n(va), n(vb)
                loss.backward()
                out[f"{tag}.loss"] = np.array(loss.item())
                for k, v in _grad_summary(model).items():
                    out[f"{tag}.{k}"] = v
        mod.query_ball_point = orig_q
    torch.manual_seed(4242)
    cs = R.replay_centres(B)
    out["centres1"], out["centres2"] = n(cs[0]).astype(np.int16), n(cs[1]).astype(np.int16)
    save("f3.npz", **out)


def golden_pt():
    """models/point_transformer.py: the reference model in eval mode (the constructor's
    dropout 0.1 is then inactive) in float32 and float64, and in train mode with every dropout probability set to 0
    with an MSE loss to a target direction (no reference script trains this model; the harness is ours)."""
    from models.point_transformer import PointTransformer
    B, N = 4, 256
    xyz, _, _, fwd = R.synthetic_clouds(B, N, seed=99)
    out = dict(xyz=n(xyz), target=n(fwd[:B]))
    for dt_name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        torch.manual_seed(42)
        model = PointTransformer()
        torch.manual_seed(5)
        with torch.no_grad():   # the clones of one encoder layer start identical; perturb them so that layers differ
            for p in model.parameters():
                p.add_(0.02 * torch.randn_like(p))
        if dt_name == "f32":
            out.update({f"pt_{k}": v for k, v in _param_checksums(model).items()})
        model = model.to(dt).eval()
        with torch.no_grad():
            y = model(xyz.to(dt))
        out[f"pt_{dt_name}.eval_out"] = n(y)
        # per-layer activations through the reference's own encoder layers (slow path: no nested tensors in train mode)
        model.train()
        for m in model.modules():
            if isinstance(m, nn.Dropout):
                m.p = 0.0
            if isinstance(m, nn.MultiheadAttention):
                m.dropout = 0.0
        x = model.input_proj(xyz.to(dt))
        for li, layer in enumerate(model.transformer.layers):
            x = layer(x)
            out[f"pt_{dt_name}.layer{li}_ck"] = np.array([x.double().sum().item(), x.double().abs().sum().item()])
        y = model(xyz.to(dt))
        loss = ((y - fwd[:B].to(dt)) ** 2).mean()
        loss.backward()
        out[f"pt_{dt_name}.train_out"], out[f"pt_{dt_name}.loss"] = n(y), np.array(loss.item())
        for k, v in _grad_summary(model).items():
            out[f"pt_{dt_name}.{k}"] = v
    save("pt.npz", **out)


def load_simple():
    """simple_pointnet_train.py as a module: its training only starts under `if __name__ == "__main__"` (:271)."""
    spec = importlib.util.spec_from_file_location("ref_simple_pointnet_train", os.path.join(REF, "simple_pointnet_train.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def golden_simple():
    """BASELINE configs[0]: the reference's own SimplePointNet (simple_pointnet_train.py:86-113) with nn.MSELoss (:243) on
    256-point clouds, batch 4 -- float32 and float64, without dropout and with an injected mask; train-mode outputs, loss,
    gradient summaries, the BatchNorm running statistics after that one forward, and the eval-mode outputs that follow."""
    ref = load_simple()
    B, N = 4, 256
    xyz, _, _, fwd = R.synthetic_clouds(B, N, seed=77)
    mask = (torch.rand(B, 128, generator=torch.Generator().manual_seed(9)) < 0.7).float()
    crit = nn.MSELoss()
    out = dict(xyz_ck=np.array([xyz.double().sum().item(), xyz.double().abs().sum().item()]), fwd=n(fwd),
               drop_mask=n(mask).astype(np.uint8))
    for dt_name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        for variant in ("nodrop", "mask"):
            torch.manual_seed(42)
            model = ref.SimplePointNet()
            if dt_name == "f32" and variant == "nodrop":
                out.update(_param_checksums(model))
            model = model.to(dt)
            model.dropout = nn.Identity() if variant == "nodrop" else _Mask(mask, 0.3)
            model.train()
            res = model(xyz.to(dt))
            loss = crit(res, fwd.to(dt))
            loss.backward()
            tag = f"{dt_name}_{variant}"
            out[f"{tag}.out"], out[f"{tag}.loss"] = n(res), np.array(loss.item())
            for k, v in _grad_summary(model).items():
                out[f"{tag}.{k}"] = v
            if variant == "nodrop":
                for k, v in model.state_dict().items():
                    if "running" in k:
                        out[f"{tag}.after.{k}"] = n(v)
                model.eval()
                with torch.no_grad():
                    out[f"{tag}.eval_out"] = n(model(xyz.to(dt)))
    save("simple.npz", **out)


def golden_vm_gt():
    """data_process/demo_vm_gt/*.txt + the mu values printed in 2d_single_peak_vM_test.ipynb."""
    d = os.path.join(REF, "data_process/demo_vm_gt")
    cases = {}
    for fn in sorted(os.listdir(d)):
        cases[fn] = open(os.path.join(d, fn), encoding="utf-8").read()
    nb = json.load(open(os.path.join(REF, "data_process/2d_single_peak_vM_test.ipynb")))
    printed = "".join("".join(o.get("text", [])) for c in nb["cells"] if c["cell_type"] == "code"
                      for o in c.get("outputs", []))
    expected = {}
    for fn in cases:
        m = re.search(r"([-+0-9.e]+) ([-+0-9.e]+)\n\[" + re.escape(fn) + r"\]", printed)
        expected[fn] = float(m.group(2)) if m else None
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "vm_gt_kat.json"), "w", encoding="utf-8") as f:
        json.dump({"files": cases, "mu": expected}, f, indent=1, ensure_ascii=False)
    print("wrote vm_gt_kat.json", expected)


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["index", "sa", "e2e", "kl", "debug_log", "vm_gt", "f3", "pt", "simple"]
    for w in which:
        globals()[f"golden_{w}"]()
