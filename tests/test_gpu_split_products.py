"""The float32 products of the grouped levels' large GEMMs, formed two ways, against the float64 oracle (GPU).

'split' (the default, csrc/gemm_wsf3_kernels.hip, gemm_wsd3_kernels.hip, gemm_mid3_kernel in gemm_mid_kernels.hip): every float32 operand is the exact sum
of three bfloat16 numbers; six of the nine bf16 x bf16 partial products -- each exact in float32 -- are accumulated in float32 on the bf16
matrix pipe, the three dropped ones are below 2^-25 of the product.  'mfma': v_mfma_f32_32x32x2_f32.  The claim tested here is that the
first is float32-class arithmetic -- inside the same gates -- not a reduced-precision mode (rounding ONE operand to bfloat16, the bf16 mode
of tests/test_gpu_bf16.py, is off by five orders of magnitude: flat gradient rel-L2 0.44 against 1e-5 here):

  * both forms of every level of the BASELINE step (configs[1]: 32 clouds x 1024 points) are held to the SAME gate against the
    float64 oracle with every discrete decision injected (1e-5 of each tensor's max-abs, tests/test_gpu_levels_routed.py);
  * what the split form costs is measured, not assumed: v_mfma_f32_32x32x16_bf16 aligns its 16 products and the accumulator to the largest
    exponent among them and DROPS what lies 2^-26 below it (tools/mfma_round.hip, profiles/round4_mfma_bf16_accumulation.txt: 1 + 12 x 2^-27
    gives 1, and 1 - 1 + 2^-30 gives 0), where the float32 instruction is a chain of correctly rounded fused multiply-adds.  With the
    leading and the small products in accumulators of their own (the forward kernels) the error is 0.8 - 0.9 x the float32 MFMA form's;
    in the backward kernels (one accumulator, or a second one for the small products where registers allow) it is 1 - 2.7 x.
    Asserted: <= 3.5 x (+ 2e-7);
  * on operands scaled over thirty orders of magnitude the two forms agree with each other to float32 rounding;
  * the wave-pair kernel's bounded LDS polls never gave up.

Reference being restated: models/pointnet_pp_8dir.py:21-43 (float32 conv -> BatchNorm -> ReLU -> max) and its autograd backward.
"""
import pytest
import torch

from conftest import relmax
from test_gpu_levels_routed import B, GATE, _run_level, net  # noqa: F401  (net: the module-scoped fixture, instantiated for this module)

pytestmark = pytest.mark.gpu

LEVELS = {"sa1": ("xyz", None, 0, "d_l1"), "sa2": ("l1_xyz", "l1", 1, "d_l2"), "sa3": ("l2_xyz", "l2", None, "d_l3")}


@pytest.fixture()
def products():
    from pnpp_hip import ops
    before = ops.get_float32_products()
    yield ops
    ops.set_float32_products(before)


def _level(oracle, net, prefix, ops, mode):
    xyz_k, pts_k, ci, d_k = LEVELS[prefix]
    getattr(net["model"], prefix).load_state_dict({k[len(prefix) + 1:]: v for k, v in net["state"].items() if k.startswith(prefix + ".")})
    ops.set_float32_products(mode)
    return _run_level(oracle, net, prefix, net[xyz_k], None if pts_k is None else net[pts_k], None if ci is None else net["centres"][ci],
                      net[d_k], ci is None)   # sa3 is the group_all level (its 512 -> 1024 forward product: gemm_mid3 / gemm_mid)


@pytest.mark.parametrize("prefix", ["sa1", "sa2", "sa3"])
def test_split_products_stay_inside_the_float32_gates(oracle, net, products, prefix):
    split = _level(oracle, net, prefix, products, "split")
    mfma = _level(oracle, net, prefix, products, "mfma")
    assert split.keys() == mfma.keys()
    print(f"\n[{prefix}] error against float64 (rel-to-max), split | mfma:\n    " +
          "\n    ".join(f"{k:20s} {split[k]:.2e} | {mfma[k]:.2e}" for k in split))
    assert max(split.values()) <= GATE and max(mfma.values()) <= GATE, (split, mfma)
    worse = {k: (split[k], mfma[k]) for k in split if split[k] > 3.5 * mfma[k] + 2e-7}
    assert not worse, worse


@pytest.mark.parametrize("scale", [1e-15, 1.0, 1e15])
def test_the_two_forms_agree_over_thirty_orders_of_magnitude(products, scale):
    """A grouped level (8 clouds, 32 centres x 32 neighbours = 8,192 rows: the smallest the split kernels take) on features scaled by
    1e-15 ... 1e15 -- BatchNorm brings every layer back to O(1), so only layer 0's operands see the scale, and its pieces must carry it
    (bf16 has float32's exponent range).  Forward output and every gradient of the two forms agree to float32 rounding."""
    from models.pointnet_pp_8dir import PointNetSetAbstraction
    torch.manual_seed(3)
    Bc, N, S, K, D = 8, 256, 32, 32, 64
    sa = PointNetSetAbstraction(S, K, D, [128, 128, 256], False).cuda().train()
    state = {k: v.clone() for k, v in sa.state_dict().items()}
    xyz = torch.randn(Bc, N, 3, device="cuda")
    pts = (torch.randn(Bc, N, D, device="cuda") * scale).requires_grad_(True)
    centres = torch.stack([torch.randperm(N)[:S] for _ in range(Bc)]).cuda()
    gy = torch.randn(Bc, S, 256, device="cuda")
    out = {}
    for mode in ("split", "mfma"):
        sa.load_state_dict(state)
        sa.zero_grad()
        pts.grad = None
        products.set_float32_products(mode)
        _, y = sa(xyz, pts, centres)
        y.backward(gy)
        torch.cuda.synchronize()
        out[mode] = {"out": y.detach().clone(), "d_points": pts.grad.detach().clone(),
                     **{"d_" + n: p.grad.detach().clone() for n, p in sa.named_parameters() if p.grad is not None and float(p.grad.abs().max()) > 0}}
    res = {k: relmax(out["split"][k], out["mfma"][k].double()) for k in out["split"]}
    print(f"\n[scale {scale:g}] split vs mfma, rel-to-max: " + ", ".join(f"{k} {v:.1e}" for k, v in res.items()))
    assert all(torch.isfinite(v).all() for v in out["split"].values())
    assert max(res.values()) <= 4e-6, res


def test_zz_wave_pair_polls_never_timed_out():
    from pnpp_hip import _lib
    torch.cuda.synchronize()
    assert _lib.lib().pnpp_debug_wsd3_timeouts() == 0
