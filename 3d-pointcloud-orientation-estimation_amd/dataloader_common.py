"""Shared PLY / sampling helpers of the three drop-in dataloader modules.

Reference: dataloader_single_peak_vonMises.py:5-14, dataloader_multi_peak_vonMises.py:6-26,
dataloader_8dir_sampled.py:6-16 (three copies of the same two functions).

The reference parses every 10k-line ASCII PLY with np.loadtxt on every __getitem__; at the rates the GPU
path sustains that is three to four orders of magnitude too slow.  read_ply therefore keeps an
optional binary side cache: set PNPP_PLY_CACHE=1 to write `<file>.ply.npy` next to the source (or
PNPP_PLY_CACHE_DIR=<dir> for a separate directory) the first time a file is parsed and memory-map it afterwards.
The parsed values are identical (float32 of the first three columns).
"""
import hashlib
import os

import numpy as np


def _cache_path(p):
    d = os.environ.get("PNPP_PLY_CACHE_DIR")
    if d:
        os.makedirs(d, exist_ok=True)
        return os.path.join(d, hashlib.sha1(os.path.abspath(str(p)).encode()).hexdigest() + ".npy")
    if os.environ.get("PNPP_PLY_CACHE") == "1":
        return str(p) + ".npy"
    return None


def _parse_ascii_ply(p):
    with open(p, "r") as f:
        while True:
            line = f.readline()
            if not line or line.strip() == "end_header":
                break
        body = f.read()
    rows = [ln.split() for ln in body.splitlines() if ln.strip()]
    if not rows:
        return np.zeros((0, 3), np.float32)
    width = min(len(r) for r in rows)
    if width < 3:
        raise ValueError(f"PLY vertex rows of {p} have fewer than three columns")
    flat = np.array([r[:3] for r in rows], dtype=np.float32)
    return flat


def read_ply(p):
    """ASCII PLY -> (n,3) float32 array of the first three vertex properties (x y z)."""
    c = _cache_path(p)
    if c and os.path.exists(c) and os.path.getmtime(c) >= os.path.getmtime(p):
        try:
            return np.load(c, mmap_mode="r")
        except Exception:      # a damaged or foreign file is a cache miss: parse the source again and replace it
            pass
    pts = _parse_ascii_ply(p)
    if c:
        # several DataLoader workers (and ranks) may want the same file at once: each writes its own temporary and
        # renames it into place -- a reader sees either no cache file or a complete one, never a partial write
        tmp = f"{c}.tmp{os.getpid()}"
        try:
            with open(tmp, "wb") as f:
                np.save(f, pts)
            os.replace(tmp, c)
        except OSError:
            try:
                os.remove(tmp)
            except OSError:
                pass
    return pts


def sample_pts(arr, num=10_000):
    """`num` points of `arr`: without replacement when enough points exist, with replacement otherwise
    (np.random global generator, as in the reference)."""
    n = len(arr)
    if n == 0:
        return np.asarray(arr)
    idx = np.random.choice(n, num, replace=n < num)
    return np.asarray(arr)[idx]


class LabelledPlyDataset:
    """What the reference's three dataset classes share (dataloader_single_peak_vonMises.py:24-52,
    dataloader_multi_peak_vonMises.py:29-86, dataloader_8dir_sampled.py:27-57): a list of sample tuples whose first entry
    is the PLY path and whose last entry is the class name, a class-name -> index map (given, or numbered in order of
    first appearance / sorted, as each reference class does), and `num_points` randomly drawn points per item.
    Subclasses supply `ground_truth(sample)` -- the columns between the cloud and the label."""

    sorted_labels = False          # how a missing label_map is built: order of first appearance, or sorted names

    def __init__(self, samples, num_points, label_map=None):
        self.samples = list(samples)
        self.num_points = num_points
        names = [s[-1] for s in self.samples]
        if not label_map:
            names = sorted(set(names)) if self.sorted_labels else list(dict.fromkeys(names))
            label_map = {n: i for i, n in enumerate(names)}
        self.label2id = label_map

    def __len__(self):
        return len(self.samples)

    def cloud(self, ply_path):
        import torch
        return torch.from_numpy(np.ascontiguousarray(sample_pts(read_ply(ply_path), self.num_points), dtype=np.float32))

    def ground_truth(self, sample):
        raise NotImplementedError

    def label(self, sample):
        return self.label2id[sample[-1]]

    def __getitem__(self, idx):
        sample = self.samples[idx]
        return (self.cloud(sample[0]), *self.ground_truth(sample), self.label(sample))


class DeviceCloudBank:
    """Full clouds resident in HBM with on-device `sample_pts` (throughput mode of the data path).

    The reference draws `np.random.choice(len, num, replace=len<num)` per item on the host every epoch
    (dataloader_single_peak_vonMises.py:12-14); at the rates the GPU path sustains that and the host-to-device copy of
    every batch are the bottleneck.  Here every cloud is uploaded once (padded to the longest, 288 GB of HBM hold
    millions of 10k-point clouds) and a batch is drawn by one kernel launch: `sample(cloud_ids, num_points)` returns
    a fresh (B, num_points, 3) float32 device tensor with the same distribution as sample_pts -- without replacement
    when the cloud has enough points, with replacement otherwise.  Draws are a pure function of (seed, draw counter).
    """
    MAX_POINTS = 1 << 24  # the kernel's key table is sized by num_points, not by the cloud (any length the bank tensor can hold)

    def __init__(self, clouds, device, seed: int = 0):
        import torch
        clouds = [np.asarray(c, dtype=np.float32).reshape(-1, 3) for c in clouds]
        if not clouds:
            raise ValueError("DeviceCloudBank: no clouds")
        longest = max(len(c) for c in clouds)
        if longest > self.MAX_POINTS:
            raise ValueError(f"DeviceCloudBank: a cloud has {longest} points, the on-device sampler takes at most "
                             f"{self.MAX_POINTS}; thin it once on the host first")
        lmax = max(longest, 1)
        host = np.zeros((len(clouds), lmax, 3), np.float32)
        for i, c in enumerate(clouds):
            host[i, :len(c)] = c
        self.bank = torch.from_numpy(host).to(device)
        self.lengths = torch.tensor([len(c) for c in clouds], dtype=torch.int32, device=device)
        self.seed, self.draws = int(seed), 0

    def __len__(self):
        return self.bank.shape[0]

    def sample(self, cloud_ids, num_points: int):
        import torch
        from pnpp_hip import ops
        ids = torch.as_tensor(cloud_ids, dtype=torch.int32, device=self.bank.device)
        self.draws += 1
        return ops.subsample_points(self.seed, self.draws, self.bank, self.lengths, num_points, ids)


class BankLoader:
    """Iterates (xyz (B,num_points,3), *targets[ids]) batches entirely on the device: the loaders' shuffle + the bank's
    on-device subsampling; drop-in for a torch DataLoader in pnpp_hip.trainer.fit / evaluate."""

    def __init__(self, bank: DeviceCloudBank, targets, num_points: int, batch: int, shuffle: bool, seed: int = 0,
                 drop_last: bool = False):
        import torch
        self.bank, self.num_points, self.batch, self.shuffle, self.drop_last = bank, num_points, batch, shuffle, drop_last
        self.targets = [t.to(bank.bank.device) for t in targets]
        self.gen = torch.Generator().manual_seed(seed)

    def __len__(self):
        n = len(self.bank)
        return n // self.batch if self.drop_last else (n + self.batch - 1) // self.batch

    def __iter__(self):
        import torch
        n = len(self.bank)
        order = torch.randperm(n, generator=self.gen) if self.shuffle else torch.arange(n)
        for i in range(0, n, self.batch):
            ids = order[i:i + self.batch]
            if self.drop_last and len(ids) < self.batch:
                break
            dev_ids = ids.to(self.bank.bank.device)
            yield (self.bank.sample(dev_ids, self.num_points), *(t[dev_ids] for t in self.targets))
