#!/usr/bin/env python3
"""Phase stamps of the fused backward GEMM kernels (`s_memtime` around every phase of wave 0 of one workgroup, a full `s_waitcnt`
at each stamp -- which serialises what would otherwise overlap, so the numbers locate costs, they do not add up to the
un-instrumented time).  Needs the instrumented build, which is not the shipped one:

    PNPP_STAMPS=1 python -c "import __graft_entry__ as g; g.build()"   (after deleting csrc/_obj/gemm_kernels.o)
    python tools/phase_stamps.py
"""
import ctypes, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3d-pointcloud-orientation-estimation_amd"))
import torch
from models.pointnet_pp_vonMises import PointNetPPVonMises
from pnpp_hip import ops, optim, _lib
torch.manual_seed(0)
m = PointNetPPVonMises(sampler="device").cuda().train()
opt = optim.FlatAdam(m.parameters())
xyz = torch.randn(32, 1024, 3, device="cuda"); mu = torch.zeros(32, device="cuda"); kap = torch.ones(32, device="cuda")
lib = _lib.lib()
lib.pnpp_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
def step():
    opt.zero_grad(); ops.vm_head_kl_loss_backward(m.features(xyz), mu, kap); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
names = ["loop top (prev tile tail) + gdm issue", "barrier 1", "stage (wait loads + transform + ds_write)", "barrier 2", "fetch issue + zp issue + dA MFMA loop", "epilogue", "barrier 3", "dW loop"]
for kd in (256, 128, 1128, 1064):
    lib.pnpp_debug_stamps(None, kd)
    N = 20
    for _ in range(N): step()
    buf = (ctypes.c_ulonglong * 16)()
    lib.pnpp_debug_stamps(buf, 0)
    tiles = N * 4
    tot = sum(buf[i] for i in range(8))
    print(f"KD={kd}")
    for i, n in enumerate(names):
        print(f"  {n:45s} {buf[i]/tiles:10.0f} ticks/tile  {100*buf[i]/max(tot,1):5.1f}%")
    print("  total per tile", tot / tiles)
    print(f"  per launch: weight panel {buf[11] / N:.0f}, constants {buf[12] / N:.0f}, first tile's loads {buf[8] / N:.0f} ticks, tail (dW partial + statistics, stores complete) {buf[10] / N:.0f} ticks")
