#!/usr/bin/env python3
"""Phase stamps of gemm_wsq_kernel / gemm_wsq2_kernel INSIDE the training step (s_memtime around the phases of wave 0 of workgroup 8, no
extra waits).  Needs a library built with -DPNPP_STAMPS on gemm_wsq_kernels.hip (tools/build_variant_src.sh qst gemm_wsq_kernels.hip
-DPNPP_STAMPS), installed as libpnpp_hip.so; PNPP_WSQ_FORM=1|2 selects the kernel."""
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
lib.pnpp_debug_wsq_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
def step():
    opt.zero_grad(); ops.vm_head_kl_loss_backward(m.features(xyz), mu, kap); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
names = ["loop turn-around", "staging / wait + fix-ups", "barrier A", "dA product", "dW product + epilogue", "barrier B", "", "", "prologue", "tail"]
lib.pnpp_debug_wsq_stamps(None, 1)
N = 20
for _ in range(N): step()
buf = (ctypes.c_ulonglong * 16)()
lib.pnpp_debug_wsq_stamps(buf, 0)
tot = sum(buf[i] for i in range(10) if i != 6)
print("form", os.environ.get("PNPP_WSQ_FORM", "default"))
for i, n in enumerate(names):
    if buf[i] and i != 6: print(f"  {n:40s} {buf[i]/N:9.0f} ticks/launch  {100*buf[i]/max(tot,1):5.1f} %")
print(f"  total {tot/N:.0f} ticks per launch")
if buf[6]: print(f"  {tot/N:.0f} shader cycles in {buf[6]/N/100:.2f} us of real time (s_memrealtime, 100 MHz): {tot/(buf[6]*10):.3f} GHz")
