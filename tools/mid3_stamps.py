#!/usr/bin/env python3
"""Phase stamps of gemm_mid3_kernel INSIDE the training step (s_memtime in wave 4 = a stager and wave 0 = a multiplier of workgroup 8).
Needs libpnpp_hip.so built with PNPP_STAMPS=1 (pnpp_hip/build.py: -DMID3_STAMPS); the clock ticks at about the shader clock here
(a 32-cycle MFMA reads 32.0 ticks: tools/mfma_valu_overlap.hip)."""
import ctypes, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3d-pointcloud-orientation-estimation_amd"))
import torch
from models.pointnet_pp_vonMises import PointNetPPVonMises
from pnpp_hip import ops, optim, _lib
torch.manual_seed(0)
m = PointNetPPVonMises(sampler="device").cuda().train()
opt = optim.FlatAdam(m.parameters())
xyz = torch.randn(32, 1024, 3, device="cuda"); mu = torch.zeros(32, device="cuda"); kap = torch.ones(32, device="cuda")
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.pnpp_debug_mid3_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
def step():
    opt.zero_grad(); ops.vm_head_kl_loss_backward(m.features(xyz), mu, kap); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
lib.pnpp_debug_mid3_stamps(None, 1)
N = 50
for _ in range(N): step()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8)()
lib.pnpp_debug_mid3_stamps(buf, 0)
names = ["prologue (to the first barrier)", "work (staging / fragments + MFMAs)", "barrier (incl. the wait for the wave's own LDS operations)", "epilogue"]
for role, tag in ((0, "stager (wave 4)"), (1, "multiplier (wave 0)")):
    row = [buf[role * 4 + i] for i in range(4)]
    tot = sum(row)
    print(f"{tag}: {tot / N:.0f} ticks per launch")
    for n, r in zip(names, row):
        print(f"    {n:60s} {r / N:8.1f} ticks  {100 * r / max(tot, 1):5.1f} %")
