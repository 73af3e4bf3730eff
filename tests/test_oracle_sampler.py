"""CPU: the restatement of the build's counter-based samplers (oracle/sampler.py) is pinned to the published Philox4x32-10
known-answer vectors (Random123 kat_vectors: Salmon, Moraes, Dror, Shaw, SC'11), and has the distribution the reference's
draws have (torch.randperm(N)[:npoint], models/pointnet_pp_8dir.py:28; np.random.choice(len, num, replace=len<num),
dataloader_single_peak_vonMises.py:12-14)."""
import numpy as np

from oracle import sampler as S


def test_philox_known_answers():
    u = np.uint64
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = S.philox4x32_10(*(u(c) for c in ctr), *key)
        assert tuple(int(g) for g in got) == want


def test_sample_random_is_an_ordered_subset_with_uniform_marginals():
    B, N, npoint = 64, 200, 50
    cnt = np.zeros(N)
    first = np.zeros(N)
    for stream in range(40):
        idx = S.sample_random(42, stream, B, N, npoint)
        assert idx.shape == (B, npoint) and idx.min() >= 0 and idx.max() < N
        assert all(len(set(r.tolist())) == npoint for r in idx)                  # without replacement
        np.add.at(cnt, idx.reshape(-1), 1)
        np.add.at(first, idx[:, 0], 1)
    exp = 40 * B * npoint / N
    chi2 = ((cnt - exp) ** 2 / exp).sum()
    assert 0.5 * (1 - npoint / N) * N < chi2 < 1.6 * (1 - npoint / N) * N, chi2   # hypergeometric marginals
    assert first.max() < 5 * 40 * B / N                                            # the ORDER is random too (first slot uniform)
    assert not np.array_equal(S.sample_random(42, 0, 2, N, npoint)[0], S.sample_random(42, 0, 2, N, npoint)[1])   # clouds differ
    assert not np.array_equal(S.sample_random(42, 0, 1, N, npoint), S.sample_random(43, 0, 1, N, npoint))         # seeds differ


def test_subsample_indices_replacement_rule():
    a = S.subsample_indices(1, 1, 0, 5000, 1024)
    assert len(set(a.tolist())) == 1024 and a.max() < 5000                        # len >= num: without replacement
    b = S.subsample_indices(1, 1, 0, 300, 1024)
    assert b.shape == (1024,) and b.max() < 300 and len(set(b.tolist())) > 250     # len < num: with replacement
    assert sorted(S.subsample_indices(1, 1, 0, 1024, 1024).tolist()) == list(range(1024))   # len == num: a permutation
    assert S.subsample_indices(1, 1, 0, 0, 8).size == 0
