#!/usr/bin/env python3
"""Phase stamps of gemm_wsp_kernel (s_memtime around the phases of wave 0 of one workgroup, a full s_waitcnt at each stamp: the
numbers locate costs, they do not add up to the un-instrumented time).  Needs a library built with -DPNPP_STAMPS on
gemm_wsp_kernels.hip (tools/build_wsp_variant.sh stamps -DPNPP_STAMPS), installed as libpnpp_hip.so."""
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
lib.pnpp_debug_wsp_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
def step():
    opt.zero_grad(); ops.vm_head_kl_loss_backward(m.features(xyz), mu, kap); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
names = ["loop top", "staging (wait loads, transform, ds_write, fix-up)", "fetch issue + dA loop (+ wait for the fetched loads)", "epilogue", "dW loop",
         "prologue: constants (per launch x 1/4)", "prologue: weight panel (x 1/4)", "7"]
lib.pnpp_debug_wsp_stamps(None, 1)
N = 20
for _ in range(N): step()
buf = (ctypes.c_ulonglong * 32)()
lib.pnpp_debug_wsp_stamps(buf, 0)
for k, kd in enumerate((64, 128)):
    b = buf[16 * k:16 * k + 16]
    strips = N * 4
    tot = sum(b[i] for i in range(8))
    print(f"KD={kd}")
    for i, n in enumerate(names):
        if b[i]: print(f"  {n:55s} {b[i]/strips:10.0f} ticks/strip  {100*b[i]/max(tot,1):5.1f}%")
    print("  total per strip", tot / strips, " prologue", b[8] / N, " tail", b[9] / N)
