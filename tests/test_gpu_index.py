"""GPU parity, index kernels: HIP (through the C ABI) vs the CPU oracle and the reference-captured
golden vectors.  Bar: bit-exact (distance matrix bit-equal, index arrays equal)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def ops():
    from pnpp_hip import ops
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return ops


def test_square_distance_bit_exact(ops, oracle, golden):
    g = golden("index.npz")
    for i in range(3):
        got = ops.square_distance(_t(g[f"sq{i}_src"]).cuda(), _t(g[f"sq{i}_dst"]).cuda()).cpu().numpy()
        assert got.tobytes() == g[f"sq{i}_out"].tobytes()
    # a larger random case against the oracle (self-distances included: slightly negative values)
    x = torch.rand(3, 700, 3) * 4 - 2
    got = ops.square_distance(x[:, :50].cuda(), x.cuda()).cpu()
    assert got.numpy().tobytes() == oracle.square_distance(x[:, :50], x).numpy().tobytes()


def test_knn_matches_reference_sets_and_oracle_order(ops, oracle, golden):
    g = golden("index.npz")
    xyz = _t(g["knn_xyz"])
    c1, c2 = _t(g["knn_c1"].astype(np.int64)), _t(g["knn_c2"].astype(np.int64))
    new1 = oracle.index_points(xyz, c1)
    idx1 = ops.knn(new1.cuda(), xyz.cuda(), 32).cpu().numpy()
    assert np.array_equal(np.sort(idx1, -1), g["knn_idx1_sorted"])            # reference's sets
    assert np.array_equal(idx1, oracle.knn_indices(new1, xyz, 32).numpy())    # oracle's order too
    new2 = oracle.index_points(new1, c2)
    idx2 = ops.knn(new2.cuda(), new1.cuda(), 32).cpu().numpy()
    assert np.array_equal(np.sort(idx2, -1), g["knn_idx2_sorted"])
    xr, cr = _t(g["knnr_xyz"]), _t(g["knnr_c"].astype(np.int64))
    idxr = ops.knn(oracle.index_points(xr, cr).cuda(), xr.cuda(), 7).cpu().numpy()
    assert np.array_equal(np.sort(idxr, -1), g["knnr_idx_sorted"])


@pytest.mark.parametrize("B,N,S,k", [(1, 1, 1, 1), (2, 33, 5, 33), (3, 1500, 9, 64), (2, 2500, 7, 128), (1, 10000, 16, 32)])
def test_knn_ragged_and_multi_tile(ops, oracle, B, N, S, k):
    g = torch.Generator().manual_seed(B * 1000 + N)
    xyz = torch.rand(B, N, 3, generator=g) * 2 - 1
    new = xyz[:, torch.randperm(N, generator=g)[:S]] if N >= S else xyz[:, :S]
    got = ops.knn(new.contiguous().cuda(), xyz.cuda(), k).cpu().numpy()
    assert np.array_equal(got, oracle.knn_indices(new, xyz, k).numpy())


def test_knn_duplicates_lowest_index_wins(ops, oracle):
    """Real clouds smaller than num_points are sampled with replacement (dataloader_*.py:12-14): exact ties."""
    base = torch.rand(1, 40, 3)
    xyz = base[:, torch.randint(0, 40, (200,))]
    got = ops.knn(xyz[:, :10].contiguous().cuda(), xyz.cuda(), 16).cpu().numpy()
    assert np.array_equal(got, oracle.knn_indices(xyz[:, :10], xyz, 16).numpy())


def test_knn_errors(ops):
    x = torch.rand(1, 5, 3).cuda()
    with pytest.raises(RuntimeError, match="out of range"):
        ops.knn(x[:, :2].contiguous(), x, 6)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.knn(x[:, :2].cpu(), x.cpu(), 2)
    with pytest.raises(TypeError):
        ops.knn(x[:, :2].double(), x.double(), 2)


def test_fps_matches_demo(ops, oracle, golden):
    g = golden("index.npz")
    got = ops.farthest_point_sample(_t(g["knn_xyz"]).cuda(), 128, _t(g["fps_start"])).cpu().numpy()
    assert np.array_equal(got, g["fps_idx"])
    got = ops.farthest_point_sample(_t(g["knnr_xyz"]).cuda(), 17, _t(g["fpsr_start"])).cpu().numpy()
    assert np.array_equal(got, g["fpsr_idx"])


@pytest.mark.parametrize("B,N,npoint", [(1, 1, 1), (2, 100, 100), (2, 10, 25),   # npoint > N: the reference repeats points, so do we
                                        (32, 1024, 128), (2, 2048, 128), (3, 4097, 40), (1, 10000, 64),
                                        (2, 16384, 128),      # the largest cloud held entirely in registers (1024 threads x 16)
                                        (1, 20000, 48)])      # registers + the LDS tail
def test_fps_sizes(ops, oracle, B, N, npoint):
    g = torch.Generator().manual_seed(N)
    xyz = torch.rand(B, N, 3, generator=g)
    start = torch.randint(0, N, (B,), generator=g)
    got = ops.farthest_point_sample(xyz.cuda(), npoint, start).cpu().numpy()
    assert np.array_equal(got, oracle.farthest_point_sample(xyz, npoint, start.numpy()).numpy())


def test_fps_duplicates_take_the_first_maximum(ops, oracle):
    """Duplicated points give exact distance ties: torch.max(distance, -1)[1] (PointNet++Demo.py:28) returns the first."""
    g = torch.Generator().manual_seed(5)
    base = torch.rand(1, 300, 3, generator=g)
    xyz = torch.cat([base, base, base[:, :100]], 1)[:, torch.randperm(700, generator=g)]
    start = torch.tensor([17])
    got = ops.farthest_point_sample(xyz.cuda(), 64, start).cpu().numpy()
    assert np.array_equal(got, oracle.farthest_point_sample(xyz, 64, start.numpy()).numpy())


def test_fps_limits(ops):
    with pytest.raises(ValueError):          # beyond registers + LDS of one CU
        ops.farthest_point_sample(torch.rand(1, 30000, 3).cuda(), 8, torch.zeros(1, dtype=torch.long))


def test_ball_query_matches_demo(ops, oracle, golden):
    g = golden("index.npz")
    xyz = _t(g["knn_xyz"])
    new = oracle.index_points(xyz, _t(g["fps_idx"].astype(np.int64)))
    for r in (0.2, 0.4):
        got = ops.ball_query(r, 32, xyz.cuda(), new.cuda()).cpu().numpy()
        assert np.array_equal(got, g[f"ball_{r}"])
    xr = _t(g["knnr_xyz"])
    newr = oracle.index_points(xr, _t(g["fpsr_idx"].astype(np.int64)))
    assert np.array_equal(ops.ball_query(0.3, 5, xr.cuda(), newr.cuda()).cpu().numpy(), g["ballr_0.3"])


def test_ball_query_empty_and_multi_tile(ops, oracle):
    xyz = torch.rand(2, 3000, 3)
    far = torch.full((2, 3, 3), 50.0)                      # nothing inside the radius -> index N everywhere
    new = torch.cat([xyz[:, 2000:2004], far], 1)
    for r, ns in ((0.05, 8), (0.3, 64)):
        got = ops.ball_query(r, ns, xyz.cuda(), new.cuda()).cpu().numpy()
        assert np.array_equal(got, oracle.ball_query(r, ns, xyz, new).numpy())
    assert (got[:, 4:] == 3000).all()


def test_sample_random_is_a_uniform_ordered_subset(ops):
    B, N, S = 64, 1024, 128
    a = ops.sample_random(42, 1, B, N, S, "cuda").cpu().numpy()
    assert a.shape == (B, S) and a.min() >= 0 and a.max() < N
    assert all(len(set(r)) == S for r in a)                            # without replacement
    assert np.array_equal(a, ops.sample_random(42, 1, B, N, S, "cuda").cpu().numpy())   # pure function of (seed, id)
    b = ops.sample_random(42, 2, B, N, S, "cuda").cpu().numpy()
    assert not np.array_equal(a, b)
    assert len({tuple(r) for r in a}) == B                             # clouds draw independently
    # marginal uniformity: each index is picked with probability S/N; chi-square over many draws
    cnt = np.zeros(N)
    for s in range(40):
        cnt += np.bincount(ops.sample_random(7, s, B, N, S, "cuda").cpu().numpy().ravel(), minlength=N)
    exp = 40 * B * S / N
    chi2 = ((cnt - exp) ** 2 / exp).sum()
    assert 800 < chi2 < 1250, chi2                                      # ~N(1023, 45); (1 - S/N) shrinks it a little
    # first position is uniform too (ordered subset, not sorted)
    first = np.concatenate([ops.sample_random(9, s, B, N, S, "cuda").cpu().numpy()[:, 0] for s in range(40)])
    assert abs(first.mean() - (N - 1) / 2) < 4 * N / np.sqrt(12 * first.size)
    # full permutation when npoint == N
    p = ops.sample_random(1, 1, 2, 200, 200, "cuda").cpu().numpy()
    assert all(sorted(r) == list(range(200)) for r in p)


def test_sample_random_device_counter(ops):
    """The graph-capturable variant: stream id read from device memory, post-incremented by the kernel itself."""
    B, N, S = 32, 1021, 128                                    # N not a multiple of 4: padded key slots never win
    c = torch.tensor([5, 0], dtype=torch.int64, device="cuda")
    for call in range(3):
        got = ops.sample_random_dev(42, c, 1 << 40, B, N, S)
        want = ops.sample_random(42, (1 << 40) + 5 + call, B, N, S, "cuda")
        assert torch.equal(got, want)
        assert c.tolist() == [5 + call + 1, 0]                 # counter bumped once per launch, ticket word back at zero
    with pytest.raises(ValueError):
        ops.sample_random_dev(42, torch.zeros(1, dtype=torch.int64, device="cuda"), 0, B, N, S)


def test_index_points_forward_backward(ops, oracle):
    g = torch.Generator().manual_seed(3)
    for C in (3, 64, 130):
        pts = torch.randn(3, 50, C, generator=g)
        idx = torch.randint(0, 50, (3, 7, 9), generator=g)
        a = pts.clone().cuda().requires_grad_(True)
        out = ops.index_points(a, idx.cuda())
        ref_in = pts.clone().double().requires_grad_(True)
        ref = oracle.index_points(ref_in, idx)
        assert torch.equal(out.cpu(), ref.float())
        gy = torch.randn(out.shape, generator=g)
        out.backward(gy.cuda())
        ref.backward(gy.double())
        assert torch.allclose(a.grad.cpu().double(), ref_in.grad, rtol=0, atol=1e-5)
    # 2-D index form (B,S)
    out = ops.index_points(pts.cuda(), idx[:, :, 0].cuda())
    assert torch.equal(out.cpu(), oracle.index_points(pts, idx[:, :, 0]))


def test_models_base_surface(oracle, golden):
    from models import base
    g = golden("index.npz")
    xyz = _t(g["knn_xyz"]).cuda()
    new = base.index_points(xyz, _t(g["knn_c1"].astype(np.int64)).cuda())
    idx = base.query_ball_point(new, xyz, 32)
    assert idx.dtype == torch.int64 and idx.shape == (2, 128, 32)
    assert np.array_equal(np.sort(idx.cpu().numpy(), -1), g["knn_idx1_sorted"])
    d = base.square_distance(new, xyz)
    assert d.shape == (2, 128, 1024)


def test_subsample_points_on_device(ops):
    """sample_pts on the device: without replacement when the cloud is large enough, with replacement otherwise."""
    import dataloader_common as dc
    rng = np.random.default_rng(0)
    lens = [5000, 300, 1024, 0, 1, 16383]
    clouds = [rng.standard_normal((n, 3)).astype(np.float32) + 10.0 * i for i, n in enumerate(lens)]
    bank = dc.DeviceCloudBank(clouds, "cuda", seed=7)
    num = 1024
    ids = torch.tensor([0, 1, 2, 3, 4, 5, 0])
    out = bank.sample(ids, num).cpu().numpy()
    assert out.shape == (7, num, 3)
    for slot, cid in enumerate(ids.tolist()):
        rows, src = out[slot], clouds[cid]
        if len(src) == 0:
            assert np.all(rows == 0)
            continue
        src_set = {r.tobytes() for r in src}
        assert all(r.tobytes() in src_set for r in rows)                       # every drawn row is a point of ITS cloud
        distinct = len({r.tobytes() for r in rows})
        if len(src) >= num:
            assert distinct == num                                             # without replacement
        else:
            assert distinct <= len(src) and (len(src) == 1 or distinct > 0.8 * min(len(src), num) * 0.6)
    assert not np.array_equal(out[0], out[6])                                  # slots draw independently
    # len == num: a permutation of the cloud
    assert {r.tobytes() for r in out[2]} == {r.tobytes() for r in clouds[2]}
    # pure function of (seed, draw counter); a new draw differs
    again = dc.DeviceCloudBank(clouds, "cuda", seed=7).sample(ids, num).cpu().numpy()
    assert np.array_equal(out, again)
    assert not np.array_equal(out, bank.sample(ids, num).cpu().numpy())
    # uniform marginals: each of the 5000 points of cloud 0 is picked with probability num/5000 per draw
    cnt = np.zeros(5000)
    lut = {r.tobytes(): i for i, r in enumerate(clouds[0])}
    big = dc.DeviceCloudBank([clouds[0]], "cuda", seed=3)
    for _ in range(30):
        rows = big.sample(torch.zeros(16, dtype=torch.int64), num).cpu().numpy().reshape(-1, 3)
        np.add.at(cnt, [lut[r.tobytes()] for r in rows], 1)
    exp = 30 * 16 * num / 5000
    chi2 = ((cnt - exp) ** 2 / exp).sum()
    assert 0.7 * 5000 < chi2 < 1.1 * 5000, chi2                                # ~ (1 - num/5000) * 4999 = 3975 +- 100
    # loader on top of it
    tgt = torch.arange(len(clouds), dtype=torch.float32)
    seen = []
    for xyz, t in dc.BankLoader(bank, [tgt], 256, batch=4, shuffle=True, seed=1):
        assert xyz.is_cuda and xyz.shape[1:] == (256, 3) and t.is_cuda
        seen += t.cpu().tolist()
    assert sorted(seen) == list(range(len(clouds)))
    # no length limit any more (the kernel's key table is sized by num, not by the cloud): a 20,000-point cloud draws distinct rows
    far = dc.DeviceCloudBank([rng.standard_normal((20000, 3)).astype(np.float32)], "cuda", seed=5)
    rows = far.sample(torch.zeros(1, dtype=torch.int64), 1024).cpu().numpy()[0]
    assert len({r.tobytes() for r in rows}) == 1024


def test_paired_device_draw_equals_two_draws():
    """pnpp_sample_random_dev2: the centre draws of two stacked levels in one launch are bit-identical to the two
    separate launches and leave the device counter where they would."""
    from pnpp_hip import ops
    for base in (0, 41):
        c_a = torch.tensor([base, 0], dtype=torch.int64, device="cuda")
        c_b = c_a.clone()
        a1 = ops.sample_random_dev(1234, c_a, 7, 5, 1000, 128)
        a2 = ops.sample_random_dev(1234, c_a, 7, 5, 128, 32)
        b1, b2 = ops.sample_random_dev2(1234, c_b, 7, 5, 1000, 128, 128, 32)
        assert torch.equal(a1, b1) and torch.equal(a2, b2)
        assert torch.equal(c_a, c_b) and int(c_b[0]) == base + 2 and int(c_b[1]) == 0


def test_device_samplers_bit_exact_vs_oracle(ops):
    """The device-side centre sampler and the on-device point subsampler against their CPU restatement
    (oracle/sampler.py, Philox4x32-10 pinned by the Random123 vectors in tests/test_oracle_sampler.py): the drawn INDICES
    are bit-exact, hence the gathered rows equal numpy's gather of the same indices -- not just "a plausible subset"."""
    from oracle import sampler as S
    import dataloader_common as dc
    seed = 0x1234_5678_9ABC_DEF1
    for stream, B, N, npoint in ((1, 4, 1024, 128), ((3 << 40) + 17, 3, 128, 32), (5, 2, 10000, 128), (9, 2, 300, 300)):
        got = ops.sample_random(seed, stream, B, N, npoint, "cuda").cpu().numpy()
        assert np.array_equal(got, S.sample_random(seed, stream, B, N, npoint)), (stream, B, N, npoint)
    # the graph-replayable form: stream id = offset + device counter, post-incremented by the kernel
    cnt = torch.tensor([41, 0], dtype=torch.int64, device="cuda")
    a = ops.sample_random_dev(seed, cnt, 7, 3, 1000, 64).cpu().numpy()
    b = ops.sample_random_dev(seed, cnt, 7, 3, 1000, 64).cpu().numpy()
    assert np.array_equal(a, S.sample_random(seed, 48, 3, 1000, 64)) and np.array_equal(b, S.sample_random(seed, 49, 3, 1000, 64))
    # point subsampling: without replacement (L >= num), with replacement (L < num), empty cloud, L == num
    rng = np.random.default_rng(0)
    lens = [5000, 300, 1024, 0, 1, 16383]
    clouds = [rng.standard_normal((n, 3)).astype(np.float32) for n in lens]
    bank = dc.DeviceCloudBank(clouds, "cuda", seed=77)
    ids = [0, 1, 2, 3, 4, 5, 0]
    num = 1024
    out = bank.sample(torch.tensor(ids), num).cpu().numpy()          # first draw of the bank: stream id 1
    for slot, cid in enumerate(ids):
        idx = S.subsample_indices(77, 1, slot, lens[cid], num)
        want = clouds[cid][idx] if lens[cid] else np.zeros((num, 3), np.float32)
        assert np.array_equal(out[slot], want), (slot, cid)
    # clouds far beyond what one workgroup's LDS could hold as a key table (the table is sized by num, not by the cloud), and
    # draws whose first cut misses: num close to the cloud's length (the cut keeps everything), tiny num (wide relative spread)
    lens2 = [60_000, 200_003, 1100, 17_000]
    clouds2 = [rng.standard_normal((n, 3)).astype(np.float32) for n in lens2]
    bank2 = dc.DeviceCloudBank(clouds2, "cuda", seed=78)
    for draw, num2 in enumerate((1024, 1024, 8), start=1):
        out2 = bank2.sample(torch.tensor([0, 1, 2, 3]), num2).cpu().numpy()
        for slot in range(4):
            idx = S.subsample_indices(78, draw, slot, lens2[slot], num2)
            assert np.array_equal(out2[slot], clouds2[slot][idx]), (draw, slot)
