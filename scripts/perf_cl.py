"""Dev tool: time the channels-last kernels at the Hybrid Demucs frequency-branch shapes (B = 64 clips), one launch at a time.
usage: python scripts/perf_cl.py [N]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from remfx_amd import clast

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
DEV = "cuda:0"
B = 256


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def report(name, ms, flop, nbytes):
    print(f"{name:58s} {ms:7.3f} ms  {flop / ms / 1e9:7.1f} TF/s  {nbytes / ms / 1e9:6.2f} TB/s ({nbytes / 1e9:.2f} GB)", flush=True)


def rnd(*shape):
    return (torch.randn(*shape, device=DEV) * 0.5).to(torch.bfloat16)


# (C of the decoder layer's input, rows A)
LAYERS = [int(v) for v in os.environ.get("PERF_CL_LAYERS", "48,96,192,384").split(",")]
for Cc, A in ((48, 512), (96, 128), (192, 32), (384, 8)):
    if Cc not in LAYERS:
        continue
    pos = N * A * B
    # decoder rewrite 3x3 + GLU
    f = clast.form_conv_glu(2 * Cc, Cc, 3, 3)
    w = torch.randn(2 * Cc, Cc, 3, 3, device=DEV) / (9 * Cc) ** 0.5
    ap = clast.pack(f, w)
    x = rnd(N, A, B, Cc)
    z = clast.empty(N, A, B, 2 * Cc, DEV)
    y = clast.empty(N, A, B, Cc, DEV)
    bias = torch.zeros(2 * Cc, device=DEV)
    ms = timed(lambda: clast.conv(f, ap, x, N, A, B, A, "glu", bias=bias, out0=z, out1=y))
    report(f"rewrite3x3+glu {Cc}->{2 * Cc} A={A} (z stored)", ms, 2.0 * pos * 2 * Cc * 9 * Cc, pos * 2.0 * (Cc + 2 * Cc + Cc))
    ms = timed(lambda: clast.conv(f, ap, x, N, A, B, A, "glu", bias=bias, out1=y))
    report(f"rewrite3x3+glu {Cc}->{2 * Cc} A={A} (inference)", ms, 2.0 * pos * 2 * Cc * 9 * Cc, pos * 2.0 * (Cc + Cc))
    # its input gradient + dgelu
    fd = clast.form_conv_dgrad(2 * Cc, Cc, 3, 3)
    apd = clast.pack(fd, w)
    dz = rnd(N, A, B, 2 * Cc)
    dx = clast.empty(N, A, B, Cc, DEV)
    dzp = clast.empty(N, A, B, Cc, DEV)
    zprev = rnd(N, A, B, Cc)
    ms = timed(lambda: clast.conv(fd, apd, dz, N, A, B, A, "dgelu", out0=dx, out1=dzp, aux0=zprev))
    report(f"rewrite3x3 dgrad+dgelu {2 * Cc}->{Cc}", ms, 2.0 * pos * 2 * Cc * 9 * Cc, pos * 2.0 * (2 * Cc + 3 * Cc))
    # conv_tr Cc -> Cc/2 (4x rows)
    Co = Cc // 2
    if Co % 16 == 0:
        ft = clast.form_convtr_s4(Cc, Co)
        wt = torch.randn(Cc, Co, 8, 1, device=DEV) / (2 * Cc) ** 0.5
        apt = clast.pack(ft, wt)
        zt = clast.empty(N, 4 * A, B, Co, DEV)
        st = clast.empty(N, 4 * A, B, Co, DEV)
        skip = rnd(N, 4 * A, B, Co)
        bt = torch.zeros(Co, device=DEV)
        ms = timed(lambda: clast.conv(ft, apt, y, N, A, B, A + 1, "gelu", bias=bt, out0=zt, out1=st, aux0=skip, OAo=4 * A))
        report(f"conv_tr+gelu+skip {Cc}->{Co} (rows {A}->{4 * A})", ms, 2.0 * pos * Cc * Co * 8, pos * 2.0 * (Cc + 4 * 3 * Co))
        ftd = clast.form_convtr_s4_dgrad(Cc, Co)
        aptd = clast.pack(ftd, wt)
        dzt = rnd(N, 4 * A, B, Co)
        dzrw = clast.empty(N, A, B, 2 * Cc, DEV)
        ms = timed(lambda: clast.conv(ftd, aptd, dzt, N, 4 * A, B, A, "dglu", out0=dzrw, aux0=z))
        report(f"conv_tr dgrad+dglu {Co}->{Cc}", ms, 2.0 * pos * Cc * Co * 8, pos * 2.0 * (4 * Co + 4 * Cc))
    # encoder conv k8s4 Cc -> 2Cc (rows A -> A/4) + gelu
    if A >= 4:
        fe = clast.form_conv_s4(2 * Cc, Cc)
        we = torch.randn(2 * Cc, Cc, 8, 1, device=DEV) / (8 * Cc) ** 0.5
        ape = clast.pack(fe, we)
        ze = clast.empty(N, A // 4, B, 2 * Cc, DEV)
        ye = clast.empty(N, A // 4, B, 2 * Cc, DEV)
        be = torch.zeros(2 * Cc, device=DEV)
        ms = timed(lambda: clast.conv(fe, ape, x, N, A, B, A // 4, "gelu", bias=be, out0=ze, out1=ye))
        report(f"enc conv k8s4+gelu {Cc}->{2 * Cc} (rows {A}->{A // 4})", ms, 2.0 * pos / 4 * 2 * Cc * Cc * 8, pos * 2.0 * (Cc + 2 * 2 * Cc / 4))
        fed = clast.form_conv_s4_dgrad(2 * Cc, Cc)
        aped = clast.pack(fed, we)
        dze = rnd(N, A // 4, B, 2 * Cc)
        dzr = clast.empty(N, A, B, 2 * Cc, DEV)
        gsk = rnd(N, A, B, Cc)
        ms = timed(lambda: clast.conv(fed, aped, dze, N, A // 4, B, A // 4 + 1, "dglu", out0=dzr, aux0=z, res=gsk, OAo=A))
        report(f"enc conv dgrad+skip+dglu {2 * Cc}->{Cc}", ms, 2.0 * pos / 4 * 2 * Cc * Cc * 8, pos * 2.0 * (2 * Cc / 4 + Cc + 4 * Cc))
    # encoder rewrite 1x1 + GLU
    f1 = clast.form_conv_glu(2 * Cc, Cc, 1, 1)
    w1 = torch.randn(2 * Cc, Cc, 1, 1, device=DEV) / Cc ** 0.5
    ap1 = clast.pack(f1, w1)
    ms = timed(lambda: clast.conv(f1, ap1, x, N, A, B, A, "glu", bias=bias, out0=z, out1=y))
    report(f"rewrite1x1+glu {Cc}->{2 * Cc}", ms, 2.0 * pos * 2 * Cc * Cc, pos * 2.0 * (Cc + 2 * Cc + Cc))
    del x, z, y, dz, dx, dzp, zprev
    torch.cuda.empty_cache()
