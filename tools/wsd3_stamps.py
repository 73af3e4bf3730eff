#!/usr/bin/env python3
"""Phase stamps of gemm_wsd3_kernel INSIDE the training step (s_memtime around the phases of the producer and the consumer wave of pair 0
of workgroup 8, no extra waits).  Needs libpnpp_hip.so built with PNPP_STAMPS=1 (pnpp_hip/build.py) and PNPP_SPLIT_PRODUCTS=1."""
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
lib.pnpp_debug_wsd3_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
def step():
    opt.zero_grad(); ops.vm_head_kl_loss_backward(m.features(xyz), mu, kap); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
lib.pnpp_debug_wsd3_stamps(None, 1)
N = 20
for _ in range(N): step()
buf = (ctypes.c_ulonglong * 32)()
lib.pnpp_debug_wsd3_stamps(buf, 0)
# the producer builds the image (K = 256: and multiplies its share of dW behind `image written`, which lands in the next chunk's first
# phase), the consumer does the products and the epilogue (phases 3 .. 7 of its row)
pnB = ["dZ and its pieces (incl. the wait for the operands)", "wait for the buffer (partner)", "image written and published"]
cnB = [None, None, None, "activation fragments", "wait for the chunk (partner)", "dA products", "transposed reads + dW products", "epilogue"]
for k, tag, pn, cn in ((0, "<128,32> sa1 last layer + sa2 middle layer", pnB, cnB), (1, "<256,32> sa2 last layer (dW split between the waves)", pnB, cnB)):
    for role, names in ((0, pn), (1, cn)):
        row = [buf[(k * 2 + role) * 8 + i] for i in range(len(names))]
        tot = sum(r for r, n in zip(row, names) if n)
        print(tag, "producer" if role == 0 else "consumer", f"total {tot / N:.0f} ticks per launch (strip loop only)")
        for i, n in enumerate(names):
            if n:
                print(f"    {n:52s} {row[i] / N:9.0f} ticks  {100 * row[i] / max(tot, 1):5.1f} %")
print("timeouts:", lib.pnpp_debug_wsd3_timeouts())
