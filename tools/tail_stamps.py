#!/usr/bin/env python3
"""Phase stamps of the multi-peak step's tail launch (library built with PNPP_STAMPS=1: python -m pnpp_hip.build):
    PNPP_STAMPS=1 python -m pnpp_hip.build --force && python tools/tail_stamps.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-pointcloud-orientation-estimation_amd")]
import torch
import torch.nn as nn
from pnpp_hip import _lib, ops
h = ctypes.CDLL(_lib.LIB_PATH)
B, K = 32, 256
torch.manual_seed(0)
heads = [nn.Linear(K, n).cuda() for n in (4, 8, 4)]
x = torch.randn(B, K, device="cuda", requires_grad=True)
vm = torch.zeros(B, 4, 3, device="cuda"); vm[:, :, 0] = torch.rand(B, 4, device="cuda") * 6 - 3; vm[:, :, 1] = 8.0; vm[:, :, 2] = 0.25
Kgt = torch.tensor([(1, 2, 4)[i % 3] for i in range(B)], device="cuda")
for _ in range(5):
    ops.mvm_heads_match_loss_backward(x, *heads, vm, Kgt, 0.7, 80.0)
buf = (ctypes.c_ulonglong * 16)()
h.pnpp_debug_tail_stamps(buf, 1)
n = 50
for _ in range(n):
    ops.mvm_heads_match_loss_backward(x, *heads, vm, Kgt, 0.7, 80.0)
h.pnpp_debug_tail_stamps(buf, 0)
names = ["staging", "o = x W^T", "head forward", "cost entries", "assign + head bwd + mean", "dW db dx"]
tot = sum(buf[i] for i in range(6))
for i, nm in enumerate(names):
    print(f"{nm:28s} {buf[i] / n:9.0f} ticks  {100.0 * buf[i] / tot:5.1f} %")
print(f"total {tot / n:.0f} ticks per launch (100 MHz s_memtime? shader clock: compare with the launch's us)")

# the single-peak tail (vm_fc_head_kl_step_kernel): stamps 8 .. 12
lin = nn.Linear(K, 2).cuda()
mu_gt, kap_gt = torch.rand(B, device="cuda") * 6 - 3, torch.full((B,), 8.0, device="cuda")
for _ in range(5):
    ops.vm_fc_head_kl_loss_backward(x, lin, mu_gt, kap_gt)
h.pnpp_debug_tail_stamps(buf, 1)
for _ in range(n):
    ops.vm_fc_head_kl_loss_backward(x, lin, mu_gt, kap_gt)
h.pnpp_debug_tail_stamps(buf, 0)
names = ["staging", "o = x W^T", "head + KL (4 chains)", "mean", "dW db dx"]
tot = sum(buf[8 + i] for i in range(5))
print("single-peak tail:")
for i, nm in enumerate(names):
    print(f"{nm:28s} {buf[8 + i] / n:9.0f} ticks  {100.0 * buf[8 + i] / tot:5.1f} %")
print(f"total {tot / n:.0f} ticks per launch")
