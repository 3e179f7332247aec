"""Dev probe: is cl_conv (LDS-DMA rings, counted vmcnt waits) bit-stable when another stream loads the memory system?  One 3 x 3 GLU
rewrite and one folded stride-4 form repeated on the main stream while a side stream runs (a) nothing, (b) HBM copy traffic,
(c) other cl_conv launches; every output compared bitwise with the first."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from remfx_amd import clast, ops

DEV = torch.device("cuda:0")
ops.set_gemm_precision("bf16")
g = torch.Generator().manual_seed(0)
side = torch.cuda.Stream(priority=-1)


def case(name, form, w, x, N, IA, IB, OA, mode, nout, **kw):
    ap = clast.pack(form, w)
    def run():
        outs = [clast.empty(N, OA, IB, c, DEV) for c in nout]
        args = dict(out0=outs[0]) if len(outs) == 1 else dict(out0=outs[0], out1=outs[1])
        clast.conv(form, ap, x, N, IA, IB, OA, mode, **args, **kw)
        return outs
    ref = [o.clone() for o in run()]
    torch.cuda.synchronize()
    hogA = torch.empty(256 << 20, device=DEV, dtype=torch.uint8)
    hogB = torch.empty_like(hogA)
    x2 = torch.randn(x.shape, generator=g).to(DEV).to(torch.bfloat16)
    for hog in ("none", "copy", "conv"):
        bad = 0
        for it in range(150):
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                if hog == "copy":
                    for _ in range(3):
                        hogB.copy_(hogA)
                elif hog == "conv":
                    for _ in range(2):
                        o2 = [clast.empty(N, OA, IB, c, DEV) for c in nout]
                        a2 = dict(out0=o2[0]) if len(o2) == 1 else dict(out0=o2[0], out1=o2[1])
                        clast.conv(form, ap, x2, N, IA, IB, OA, mode, **a2, **kw)
            outs = run()
            main.wait_stream(side)
            torch.cuda.synchronize()
            if any(not torch.equal(a, b) for a, b in zip(outs, ref)):
                bad += 1
        print(f"{name}: side stream {hog}: {bad} / 150 differ", flush=True)


N, A, T = 16, 32, 256
C = 192
x = torch.randn(N, A, T, C, generator=g).to(DEV).to(torch.bfloat16)
w = (torch.randn(2 * C, C, 3, 3, generator=g) / (9 * C) ** 0.5).to(DEV)
case("3x3 GLU rewrite C=192", clast.form_conv_glu(2 * C, C, 3, 3), w, x, N, A, T, A, "glu", (2 * C, C))
# the time branch's stride-4 convolution through the folded view: (N, 1, L/4, 4 C) -> (N, 1, L/4, Cout)
L4 = 4096
xf = torch.randn(8, 1, L4, 4 * C, generator=g).to(DEV).to(torch.bfloat16)
w4 = (torch.randn(2 * C, C, 8, generator=g) / (8 * C) ** 0.5).to(DEV)
case("stride-4 fold C=192 -> 384", clast.form_conv_s4_fold(2 * C, C), w4, xf, 8, 1, L4, 1, "store", (2 * C,))
