import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/3d-pointcloud-orientation-estimation_amd')
import torch
from models.pointnet_pp_8dir import PointNetSetAbstraction
from oracle import restatement as R
def rel(a, b): return float((a.double()-b.double()).norm() / b.double().norm())
for B in (8, 32):
    torch.manual_seed(1)
    sa = PointNetSetAbstraction(None, None, 256, [256, 512, 1024], group_all=True).cuda().train()
    xyz = torch.rand(B, 32, 3); pts = torch.randn(B, 32, 256).abs()
    pg = pts.cuda().requires_grad_(True)
    _, y = sa(xyz.cuda(), pg)
    gy = torch.randn(y.shape); y.backward(gy.cuda())
    P = {"sa." + k: (v.detach().cpu().double().requires_grad_(True) if "running" not in k else v.detach().cpu().double()) for k, v in sa.state_dict().items() if v.is_floating_point()}
    p64 = pts.double().requires_grad_(True)
    _, yr, _ = R.sa_forward(xyz, p64, P, "sa", None, None, True, True, None)
    (yr * gy.double()).sum().backward()
    print(B, 'fwd', rel(y.detach().cpu(), yr.detach()), 'dpts', rel(pg.grad.cpu(), p64.grad),
          [(n, round(rel(p.grad.cpu().flatten(), P['sa.'+n].grad.flatten()), 7)) for n, p in sa.named_parameters() if 'convs' not in n or 'weight' in n])
