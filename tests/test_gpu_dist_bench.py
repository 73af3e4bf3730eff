"""GPU: the real N > 1 bench step, as the driver's SCALE run would start it.

`python bench.py --gpus 2` is started as a FRESH child process (never an exec of this pytest process, which has touched the
GPU); the child launches its own two ranks.  RCCL refuses two ranks on one device, so on the one-GPU test box the ranks share
the card and exchange gradients through gloo (PNPP_DIST_BACKEND=gloo): everything else -- build_step's world > 1 branch, the
two captured graphs, the schedule trial, the collective between the replays, the fused Adam with grad_scale = 1/world -- is
the code that runs on 8 GPUs (reference step being wrapped: train_single_peak_vonMises_KL.py:77-86).
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def _run_bench(extra, env_extra=None, timeout=600):
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"PNPP_DIST_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--no-bf16-variant", "--no-roofline", *extra],
                       capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0]), r.stderr


def test_two_rank_bench_runs_the_captured_data_parallel_step():
    out, err = _run_bench([])
    assert "capture failed" not in err, err[-2000:]
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["steps"] == 3
    assert out["value"] > 0 and out["final_loss"] == out["final_loss"]
    cfg = out["config"]
    assert cfg["parallelism"] == "dp2" and cfg["global_batch"] == 64
    # both schedules were built and timed, and one was chosen by measurement
    assert "dp_schedule" in cfg and cfg["dp_schedule"]["chosen"] in cfg["dp_schedule"]["candidates_ms"]
    assert "hipGraph" in cfg["launch"]
    assert out["allreduce_exposed_us"] is not None


@pytest.mark.parametrize("no_overlap", ["0", "1"])
def test_two_rank_bench_forced_schedules(no_overlap):
    """PNPP_NO_OVERLAP pins the schedule: 0 = two graphs with the overlapped all-reduce, 1 = one graph + one all-reduce."""
    out, err = _run_bench([], {"PNPP_NO_OVERLAP": no_overlap})
    assert "capture failed" not in err, err[-2000:]
    assert out["n_ranks_seen"] == 2
    assert ("two hipGraphs" in out["config"]["launch"]) == (no_overlap == "0")


_CAPTURED_RCCL_CHILD = r'''
import os, sys, copy, json
import torch, torch.distributed as dist
ROOT = sys.argv[1]
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-pointcloud-orientation-estimation_amd")]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2], RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)          # RCCL with one rank: a real communicator, real launches
from models.pointnet_pp_vonMises import PointNetPPVonMises
from pnpp_hip import ops, optim, sampling, dist as pdist
from pnpp_hip.graph import GraphedSplitStep, GraphedStep
import synthetic
torch.manual_seed(42)
m1 = PointNetPPVonMises(sampler="device").cuda().train()
m2 = copy.deepcopy(m1)
o1, o2 = optim.FlatAdam(m1.parameters()), optim.FlatAdam(m2.parameters())
xyz, mu, kappa, _ = synthetic.rotated_clouds(8, 1024, seed=3)
xyz, mu, kappa = xyz.cuda(), mu.cuda(), kappa.cuda()
ar = lambda t: pdist.all_reduce_flat_grad(t, async_op=True)
def stage1(x, m, k):
    a, b = m1.sa1(x, None)
    return m1.sa2(a, b)
def stage2(a, b):
    _, l3 = m1.sa3(a, b)
    f = ops.fc_block(l3.view(l3.size(0), -1), m1.fc1, m1.bn1, relu=True, training=True)
    f = ops.fc_block(f, m1.fc2, m1.bn2, relu=True, dropout=m1.drop, training=True)
    return ops.vm_fc_head_kl_loss_backward(f, m1.fc3, mu, kappa)
tail = o1.offset_of(next(m1.sa3.parameters()))
res = {}
g = GraphedSplitStep(o1, stage1, stage2, [xyz, mu, kappa], tail, adopt_inputs=True, captured_all_reduce=ar)
assert g.captured_collective and g.graph2 is None
snap = sampling.snapshot()      # the device-side counters exist once a forward pass has run
for p, q in zip(m1.buffers(), m2.buffers()):
    p.copy_(q)
sampling.restore(snap)
l1 = float(g(xyz, mu, kappa))
sampling.restore(snap)
o2.zero_grad()
l2 = float(ops.vm_fc_head_kl_loss_backward(m2.trunk(xyz), m2.fc3, mu, kappa))
torch.cuda.synchronize()
res["split_loss_equal"] = l1 == l2
res["split_grad_equal"] = bool(torch.equal(o1.flat_g, o2.flat_g))       # world = 1: the sum over ranks is the gradient itself
g1 = GraphedStep(o1, lambda x, m, k: ops.vm_fc_head_kl_loss_backward(m1.trunk(x), m1.fc3, m, k), [xyz, mu, kappa], adopt_inputs=True,
                 captured_all_reduce=ar)
for p, q in zip(m1.buffers(), m2.buffers()):
    p.copy_(q)
sampling.restore(snap)
l3 = float(g1(xyz, mu, kappa))
torch.cuda.synchronize()
res["single_loss_equal"] = l3 == l2
res["single_grad_equal"] = bool(torch.equal(o1.flat_g, o2.flat_g))
print("RESULT " + json.dumps(res))
dist.destroy_process_group()
'''


def test_rccl_all_reduce_captured_inside_the_step_graph():
    """The schedules `captured_overlap` / `captured_single` of bench.py: RCCL's launches recorded as nodes of the step's hipGraph
    (forked beside the sa2 / sa1 backward pass, or at the end).  One rank is all a one-GPU box admits, but the communicator, its
    stream, the fork / join events and the capture are the real ones; the replay must reproduce the eager gradient bit for bit."""
    import socket
    from conftest import ROOT
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, "-c", _CAPTURED_RCCL_CHILD, ROOT, str(port)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][0][7:])
    assert all(res.values()), res
