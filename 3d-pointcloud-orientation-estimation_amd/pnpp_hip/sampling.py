"""Device-side centre sampling (throughput mode of models/pointnet_pp_8dir.py:28).

`torch.manual_seed(s)` still controls the draw: the kernel's Philox key is the CPU generator's initial seed, its
counter is (rank << 40) + a call counter.  The call counter lives in DEVICE memory and is post-incremented by the
sampling kernel itself, so a step captured into a hipGraph draws fresh, reproducible centres on every replay and
every rank under data parallelism gets an independent stream -- all without touching the host generator."""
import torch

from . import ops

_state = {"rank": 0, "counters": {}}


def set_rank(rank: int) -> None:
    _state["rank"] = int(rank)


def reset(calls: int = 0) -> None:
    for c in _state["counters"].values():
        c.copy_(torch.tensor([int(calls), 0], dtype=torch.int64))   # [call counter, ticket word of the kernel]


def snapshot():
    """Values of every device-side random-stream counter of this process (centre sampler, in-kernel dropout): taken before
    something that draws without being part of the run (graph warm-up / capture passes) ..."""
    return [(c, c.clone()) for c in list(_state["counters"].values()) + list(ops._dropout_counters.values())]


def restore(snap) -> None:
    """... and put back afterwards, so that a run stays a function of its seed whether or not it was captured."""
    for c, v in snap:
        c.copy_(v)


def _counter(device) -> torch.Tensor:
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    c = _state["counters"].get(key)
    if c is None:
        c = torch.zeros(2, dtype=torch.int64, device=device)         # [call counter, ticket word of the kernel]
        _state["counters"][key] = c
    return c


def device_random_centres(B: int, N: int, npoint: int, device) -> torch.Tensor:
    # stream id = (rank << 40) + 1 + calls so far; the kernel bumps the counter after reading it
    return ops.sample_random_dev(torch.initial_seed(), _counter(device), (_state["rank"] << 40) + 1, B, N, npoint)



def device_random_centres_pair(B: int, N1: int, npoint1: int, N2: int, npoint2: int, device):
    """The draws of two stacked levels (sa1 from the cloud, sa2 from sa1's centres) in one launch; the same centres as
    device_random_centres(B, N1, npoint1) followed by device_random_centres(B, N2, npoint2)."""
    return ops.sample_random_dev2(torch.initial_seed(), _counter(device), (_state["rank"] << 40) + 1, B, N1, npoint1, N2, npoint2)


class CentreRing:
    """The centres of the NEXT forward pass, drawn one step ahead (throughput mode of a captured training loop).

    The two draws of a step (sa1 from the cloud, sa2 from sa1's centres; models/pointnet_pp_8dir.py:28 twice) depend on nothing
    but the device-side call counter, so they need not open the step: `prime()` draws the first pair, and from then on the
    step's tail launch (ops.vm_fc_head_kl_loss_backward(..., next_centres=ring.job())) draws the following pair in the CUs its
    single workgroup leaves idle.  The model reads `ring.centres` instead of sampling (BackboneBNHead.use_presampled).  Same
    kernels' arithmetic, same counter sequence: the centres of step t are those the per-step draw would have produced."""

    def __init__(self, B: int, N: int, npoint1: int, npoint2: int, device):
        self.B, self.N1, self.N2 = int(B), int(N), int(npoint1)
        self.c1 = torch.empty(B, npoint1, device=device, dtype=torch.int32)
        self.c2 = torch.empty(B, npoint2, device=device, dtype=torch.int32)
        self.device = torch.device(device)
        self.primed = False

    @property
    def centres(self):
        if not self.primed:
            self.prime()
        return self.c1, self.c2

    def prime(self) -> None:
        """Fills the buffers with the draw of the current counter value (one launch; the tail keeps them filled afterwards)."""
        from . import _lib as L
        L.check(L.lib().pnpp_sample_random_dev2(torch.initial_seed() & (2**64 - 1), _counter(self.device).data_ptr(),
                                                (_state["rank"] << 40) + 1, self.B, self.N1, self.c1.shape[1], self.c1.data_ptr(),
                                                self.N2, self.c2.shape[1], self.c2.data_ptr(), ops._stream()))
        self.primed = True

    def job(self):
        """Arguments of the draw that refills the buffers (for ops.vm_fc_head_kl_loss_backward(next_centres=))."""
        return (torch.initial_seed(), _counter(self.device), (_state["rank"] << 40) + 1, self.B, self.N1, self.c1, self.N2, self.c2)
