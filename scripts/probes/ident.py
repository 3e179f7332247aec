import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from remfx_amd import losses
DEV = "cuda:0"; CLIP = 262144
g = torch.Generator().manual_seed(4)
x = (torch.randn(16, 1, CLIP, generator=g) * 0.2).to(DEV)
xg = x.clone().requires_grad_(True)
mr = losses.MultiResolutionSTFTLoss()(xg, x)
mr.backward()
print("identity loss", float(mr), "grad max", float(xg.grad.abs().max()), "grad rms", float(xg.grad.pow(2).mean().sqrt()))
y = x + 0.05 * torch.randn(16, 1, CLIP, generator=g).to(DEV)
xg2 = y.clone().requires_grad_(True)
m2 = losses.MultiResolutionSTFTLoss()(xg2, x)
m2.backward()
print("perturbed loss", float(m2), "grad max", float(xg2.grad.abs().max()), "grad rms", float(xg2.grad.pow(2).mean().sqrt()))
