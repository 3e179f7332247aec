"""Dev tool: time the channels-last weight-gradient kernel (csrc/cl_wgrad.hip) at the Hybrid Demucs frequency-branch shapes.
usage: python scripts/perf_clw.py [N]"""
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


def report(name, ms, flop, nbytes, f):
    print(f"{name:44s} {ms:7.3f} ms {flop / ms / 1e9:7.1f} TF/s {nbytes / ms / 1e9:6.2f} TB/s  [RW {f.RW} CW {f.CW} WK {f.WK} DT {f.DT} ahead {f.ahead}]",
          flush=True)


def rnd(*shape):
    return (torch.randn(*shape, device=DEV) * 0.5).to(torch.bfloat16)


LAYERS = [int(v) for v in os.environ.get("PERF_CL_LAYERS", "48,96,192,384").split(",")]
for Cc, A in ((48, 512), (96, 128), (192, 32), (384, 8)):
    if Cc not in LAYERS:
        continue
    pos = N * A * B
    x = rnd(N, A, B, Cc)
    dz = rnd(N, A, B, 2 * Cc)
    f = clast.wform_conv(2 * Cc, Cc, 3, 3)
    dw = torch.zeros(2 * Cc, Cc, 3, 3, device=DEV)
    db = torch.zeros(2 * Cc, device=DEV)
    ms = timed(lambda: clast.wgrad(f, dz, x, N, A, A, B, dw, db))
    report(f"wgrad 3x3 {Cc}->{2 * Cc} A={A}", ms, 2.0 * pos * 2 * Cc * 9 * Cc, pos * 2.0 * 3 * Cc, f)
    f1 = clast.wform_conv(2 * Cc, Cc, 1, 1)
    dw1 = torch.zeros(2 * Cc, Cc, 1, 1, device=DEV)
    ms = timed(lambda: clast.wgrad(f1, dz, x, N, A, A, B, dw1, db))
    report(f"wgrad 1x1 {Cc}->{2 * Cc} A={A}", ms, 2.0 * pos * 2 * Cc * Cc, pos * 2.0 * 3 * Cc, f1)
    if A >= 4:
        fe = clast.wform_conv_s4(2 * Cc, Cc)
        dze = rnd(N, A // 4, B, 2 * Cc)
        dwe = torch.zeros(2 * Cc, Cc, 8, 1, device=DEV)
        ms = timed(lambda: clast.wgrad(fe, dze, x, N, A // 4, A, B, dwe, db))
        report(f"wgrad k8s4 {Cc}->{2 * Cc} rows {A}->{A // 4}", ms, 2.0 * pos / 4 * 2 * Cc * Cc * 8, pos * 2.0 * (Cc + 2 * Cc / 4), fe)
    Co = Cc // 2
    if Co % 16 == 0:
        ft = clast.wform_convtr_s4(Cc, Co)
        dzt = rnd(N, 4 * A, B, Co)
        dwt = torch.zeros(Cc, Co, 8, 1, device=DEV)
        ms = timed(lambda: clast.wgrad(ft, x, dzt, N, A, 4 * A, B, dwt))
        report(f"wgrad conv_tr {Cc}->{Co} rows {A}->{4 * A}", ms, 2.0 * pos * Cc * Co * 8, pos * 2.0 * (Cc + 4 * Co), ft)
    del x, dz
    torch.cuda.empty_cache()
