"""CPU restatement (numpy) of the build's OWN counter-based samplers -- test infrastructure, like the rest of oracle/.

The reference draws centres with torch.randperm on the host generator (models/pointnet_pp_8dir.py:28) and subsamples points
with np.random.choice (dataloader_single_peak_vonMises.py:12-14); the throughput path of the build draws the same
DISTRIBUTIONS on the device from Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11;
pinned below by the Random123 known-answer vectors), so that a draw is a pure function of (seed, stream id, cloud, index):

  centre sampling   key(n) = philox(counter = (n, cloud, stream_lo, stream_hi), key = (seed_lo, seed_hi))[0];
                    the npoint points with the smallest (key, n) in ascending order -- a uniform random ordered subset
  point subsampling L >= num: the same with n over the cloud's L points; L < num: index_j = floor(key(j) * L / 2^32)
                    (with replacement), j = 0..num-1;  L == 0: zeros

Only tests/ may import this module."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32 with 10 rounds; counters are uint32 arrays (broadcast together), key two python ints.
    Returns the four output words."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & MASK for c in np.broadcast_arrays(c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)
        n1 = p1 & MASK
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
        n3 = p0 & MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def _keys(count, cloud, seed, stream_id):
    n = np.arange(count, dtype=np.uint64)
    return philox4x32_10(n, np.uint64(cloud), np.uint64(stream_id & 0xFFFFFFFF), np.uint64((stream_id >> 32) & 0xFFFFFFFF),
                         seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)[0].astype(np.uint64)


def sample_random(seed: int, stream_id: int, B: int, N: int, npoint: int) -> np.ndarray:
    """(B, npoint) int32: per cloud the npoint indices with the smallest (key, index), in that order."""
    out = np.empty((B, npoint), np.int32)
    for b in range(B):
        word = (_keys(N, b, seed, stream_id) << np.uint64(32)) | np.arange(N, dtype=np.uint64)
        out[b] = (np.sort(word)[:npoint] & MASK).astype(np.int32)
    return out


def subsample_indices(seed: int, stream_id: int, slot: int, L: int, num: int) -> np.ndarray:
    """Indices into a cloud of L points drawn for batch slot `slot` (the Philox counter uses the SLOT, not the cloud id)."""
    if L <= 0:
        return np.zeros((0,), np.int64)
    if L < num:
        u = _keys(num, slot, seed, stream_id)
        return ((u * np.uint64(L)) >> np.uint64(32)).astype(np.int64)
    word = (_keys(L, slot, seed, stream_id) << np.uint64(32)) | np.arange(L, dtype=np.uint64)
    return (np.sort(word)[:num] & MASK).astype(np.int64)
