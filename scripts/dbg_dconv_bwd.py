"""Dev: where does the fused DConv backward's dx differ from the layer-by-layer one?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from remfx_amd import nnops, ops, hdemucs
ops.set_gemm_precision("bf16")
dev = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
torch.manual_seed(1)
net = hdemucs._DConv(48, compress=4, depth=2, init=0.3, attn=False, lstm=False).to(dev)
g = torch.Generator().manual_seed(2)
x = torch.randn(N, 48, 256, generator=g).to(dev)
gy = torch.randn(N, 48, 256, generator=g).to(dev)
res = {}
for fb in (True, False, True, True):
    nnops.DCONV_FUSED_BWD = fb
    xd = x.clone().requires_grad_(True)
    y = net(xd)
    (dx,) = torch.autograd.grad(y, xd, gy)
    torch.cuda.synchronize()
    res.setdefault(fb, []).append(dx - gy)
ref = res[False][0]
for k, d in enumerate(res[True]):
    err = (d - ref).abs()
    bad = err > 0.05 * ref.abs().max()
    print(f"run {k}: rel err {float((d - ref).norm() / ref.norm()):.3e}; bad elements {int(bad.sum())}")
    if bad.any():
        idx = bad.nonzero()
        ns = idx[:, 0].unique()
        print("  bad samples", ns[:20].tolist(), "count", len(ns))
        n0 = int(ns[0])
        sub = idx[idx[:, 0] == n0]
        print("  sample", n0, "channels", sub[:, 1].unique().tolist()[:48])
        ts = sub[:, 2].unique().tolist()
        print("  positions", ts[:80], "count", len(ts))
