import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/3d-pointcloud-orientation-estimation_amd')
import torch, math
from models.pointnet_pp_vonMises import PointNetPPVonMises
from pnpp_hip import ops
from oracle import restatement as R
B = 32
torch.manual_seed(42)
model = PointNetPPVonMises()
state = {k: v.clone() for k, v in model.state_dict().items()}
model = model.cuda().train()
xyz, mu_gt, kappa_gt, _ = R.synthetic_clouds(B, 1024, seed=1234)
torch.manual_seed(4242)
centres = R.replay_centres(B)
mask = (torch.rand(B, 256, generator=torch.Generator().manual_seed(8)) < 0.5).to(torch.uint8)
P64 = R.cast_params(state, torch.float64)
mu64, kap64 = R.vonmises_forward(xyz, P64, centres, mask.float(), True, None)
R.kl_single(mu64, kap64, mu_gt.double(), kappa_gt.double()).mean().backward()
grads = []
for rep in range(3):
    model.zero_grad()
    mu, kappa = model(xyz.cuda(), centres=[c.cuda() for c in centres], drop_mask=mask.cuda())
    loss = ops.kl_von_mises_single(mu, kappa, mu_gt.cuda(), kappa_gt.cuda()).mean()
    loss.backward()
    g = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    grads.append(g)
    worst = []
    for n, p in model.named_parameters():
        ref = P64[n].grad.reshape(p.shape)
        if float(ref.abs().max()) < 1e-9: continue
        e = float((p.grad.cpu().double() - ref).norm() / ref.norm())
        worst.append((e, n))
    worst.sort(reverse=True)
    print(rep, loss.item(), worst[:6])
for n in grads[0]:
    if not torch.equal(grads[0][n], grads[1][n]) or not torch.equal(grads[0][n], grads[2][n]):
        print('NONDET', n, float((grads[0][n]-grads[1][n]).abs().max()))
