import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keypointnerf_amd import ops
R, S = 65536, 128
torch.manual_seed(17)
q = torch.rand(1, R, S, 5, device="cuda")
q[..., 0] = torch.relu(torch.randn(1, R, S, device="cuda")) * 4
zz = (torch.rand(1, R, S, device="cuda") * 0.05 + 0.005).cumsum(-1) + 2.0
q1 = q.clone().requires_grad_(True)
c1 = ops.rgba2out(q1, zz)[0]
w = torch.randn_like(c1)
(g1,) = torch.autograd.grad(c1, q1, w)
q2 = q.double().clone().requires_grad_(True)
z2 = zz.double()
dist = torch.cat([z2[..., 1:] - z2[..., :-1], 1e10 * torch.ones_like(z2[..., :1])], -1)
a = 1.0 - torch.exp(-q2[..., 0] * dist)
cw = a * torch.cumprod(torch.cat([torch.ones_like(a[..., :1]), 1 - a[..., :-1]], -1), -1)
c2 = (q2[..., 2:] * cw[..., None]).sum(-2)
(g2,) = torch.autograd.grad(c2, q2, w.double())
q3 = q.clone().requires_grad_(True)
dist3 = torch.cat([zz[..., 1:] - zz[..., :-1], 1e10 * torch.ones_like(zz[..., :1])], -1)
a3 = 1.0 - torch.exp(-q3[..., 0] * dist3)
cw3 = a3 * torch.cumprod(torch.cat([torch.ones_like(a3[..., :1]), 1 - a3[..., :-1]], -1), -1)
c3 = (q3[..., 2:] * cw3[..., None]).sum(-2)
(g3,) = torch.autograd.grad(c3, q3, w)
print("torch fp32 at the ray 57166:", g3[0, 57166, 127, 0].item(), "kernel", g1[0, 57166, 127, 0].item(), "fp64", g2[0, 57166, 127, 0].item())
af = a3[0, 57166].detach(); print("small 1-a factors:", sorted((1 - af).tolist())[:6])
ok = torch.isfinite(g2).all(-1).all(-1) & ((1 - a[..., :-1]).detach().amin(-1) > 1e-3)
print("ok frac", ok.float().mean().item())
scale = g2[ok].abs().amax((-1, -2), keepdim=True)
err = (g1[ok].double() - g2[ok]).abs() / (2e-4 * scale + 1e-6)
print("max ratio", err.max().item())
idx = (err == err.max()).nonzero()[0]
print("at", idx.tolist(), "g1", g1[ok][idx[0], idx[1], idx[2]].item(), "g2", g2[ok][idx[0], idx[1], idx[2]].item(), "scale", scale[idx[0]].item())
r = idx[0].item()
print("ray g1[:,0]", g1[ok][r, :, 0][-8:].tolist())
print("ray g2[:,0]", g2[ok][r, :, 0][-8:].tolist())
print("bad entries", (err > 1).sum().item(), "rays", (err > 1).any(-1).any(-1).sum().item(), "which cols", (err > 1).any(0).any(0).tolist(), "sample idx hist", (err > 1).any(-1).any(0).nonzero().flatten().tolist()[:40])
