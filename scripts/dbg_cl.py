import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from remfx_amd import clast
DEV = "cuda:0"
def _r(t): return t.to(torch.bfloat16).to(torch.float64)
def _cl(x): return x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)
def _cm(x): return x.detach().cpu().to(torch.float64).permute(0, 3, 1, 2).contiguous()
g = torch.Generator().manual_seed(2)
B = 256
for (Cred, M, A, N) in [(96, 192, 2, 1), (192, 192, 2, 1), (384, 192, 2, 1), (384, 192, 1, 1), (384, 96, 2, 1), (384, 384, 2, 1)]:
    dz = torch.randn(N, Cred, A, B, generator=g)
    w = torch.randn(Cred, M, 3, 3, generator=g) / (Cred * 9) ** 0.5
    form = clast.form_conv_dgrad(Cred, M, 3, 3)
    ap = clast.pack(form, w.to(DEV))
    dx = clast.empty(N, A, B, M, DEV)
    clast.conv(form, ap, _cl(dz), N, A, B, A, "store", out0=dx)
    torch.cuda.synchronize()
    ref = F.conv_transpose2d(_r(dz), _r(w), padding=1)
    got = _cm(dx)
    err = (got - ref).abs()
    print(f"Cred={Cred} M={M} A={A} BM={form.BM} NCH={form.NCH}: max abs err {float(err.max()):.3e} (ref rms {float(ref.pow(2).mean().sqrt()):.3f})")
    if float(err.max()) > 0.1:
        e = err[0]                                   # (M, A, B)
        print("  by 32-row tile:", [f"{float(e[i*32:(i+1)*32].max()):.2f}" for i in range(M // 32)])
        print("  by row a:", [f"{float(e[:, a].max()):.2f}" for a in range(A)])
        print("  by 32-pos tile:", [f"{float(e[:, :, i*32:(i+1)*32].max()):.2f}" for i in range(8)])
        # ratio test: is got a partial sum? correlate
        num = float((got * ref).sum()); den = float((ref * ref).sum())
        print("  projection of got on ref:", num / den)
