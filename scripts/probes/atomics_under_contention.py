"""Dev probe: fp64 atomicAdd reductions (rfx_row_moments: 3 - 16 workgroups per row add into one pair of doubles) on the main stream
while a side stream runs cl_conv launches: a lost update would move a mean by percent, a reordering by 1e-16."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from remfx_amd import clast, nnops, ops

DEV = torch.device("cuda:0")
ops.set_gemm_precision("bf16")
g = torch.Generator().manual_seed(0)
side = torch.cuda.Stream(priority=-1)
C = 192
xc = torch.randn(8, 1, 4096, 4 * C, generator=g).to(DEV).to(torch.bfloat16)
w4 = (torch.randn(2 * C, C, 8, generator=g) / (8 * C) ** 0.5).to(DEV)
form = clast.form_conv_s4_fold(2 * C, C)
ap = clast.pack(form, w4)
for R, L in ((256, 192 * 256), (2048, 48 * 256), (8, 262144)):
    x = torch.randn(R, L, generator=g).to(DEV)
    y0, m0, s0 = nnops.row_standardize(x, 1e-5)
    torch.cuda.synchronize()
    bad = 0
    for it in range(200):
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            for _ in range(3):
                o = clast.empty(8, 1, 4096, 2 * C, DEV)
                clast.conv(form, ap, xc, 8, 1, 4096, 1, "store", out0=o)
        y, m, s = nnops.row_standardize(x, 1e-5)
        main.wait_stream(side)
        torch.cuda.synchronize()
        dm = float(((m - m0).abs() / (s0 + 1e-30)).max())
        ds = float(((s - s0).abs() / (s0 + 1e-30)).max())
        if dm > 1e-6 or ds > 1e-6:
            bad += 1
            if bad <= 3:
                print(f"R={R} L={L} it {it}: mean moved {dm:.2e} std, std moved {ds:.2e} relative", flush=True)
    print(f"R={R} L={L}: {bad} / 200 bad")
