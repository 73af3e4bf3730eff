#!/usr/bin/env python3
"""Phase stamps of gemm_wsf_kernel INSIDE the training step (s_memtime around the phases of wave 0 of workgroup 8, no extra waits).
Needs a library built with -DPNPP_STAMPS on gemm_wsf_kernels.hip (tools/build_variant_src.sh fst gemm_wsf_kernels.hip -DPNPP_STAMPS),
installed as libpnpp_hip.so."""
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
lib.pnpp_debug_wsf_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
def step():
    opt.zero_grad(); ops.vm_head_kl_loss_backward(m.features(xyz), mu, kap); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
lib.pnpp_debug_wsf_stamps(None, 1)
N = 20
for _ in range(N): step()
buf = (ctypes.c_ulonglong * 32)()
lib.pnpp_debug_wsf_stamps(buf, 0)
names = ["prologue", "staging (+ wait for the strip)", "product", "epilogue (stores, stats, pool)", "tail"]
for w, tag in ((1, "<64,2> N=128 (sa1 L2)"), (2, "<128,2> N=128 (sa2 L1)"), (3, "<128,2> N=256 (sa2 L2)")):
    row = [buf[w * 8 + i] for i in range(5)]
    tot = sum(row)
    print(tag, f"total {tot / N:.0f} ticks per launch")
    for i, n in enumerate(names):
        print(f"    {n:34s} {row[i] / N:9.0f} ticks  {100 * row[i] / max(tot, 1):5.1f} %")
