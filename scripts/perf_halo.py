"""Dev tool: same-process A/B of the halo-tile kernel against the tap-major kernel on the Hybrid Demucs layers it takes over
(B = 64): forward (conv + GLU store), input gradient, per layer, HIP-event timed.   python scripts/perf_halo.py [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from remfx_amd import convplan, ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
ops.set_gemm_precision("bf16")
LAYERS = [  # name, N, C, (A, B), kernel, pad, dil, glu
    ("dec rewrite 48->96 3x3 512x256", 64, 48, (512, 256), (3, 3), (1, 1), (1, 1), True),
    ("dec rewrite 96->192 3x3 128x256", 64, 96, (128, 256), (3, 3), (1, 1), (1, 1), True),
    ("dec rewrite 192->384 3x3 32x256", 64, 192, (32, 256), (3, 3), (1, 1), (1, 1), True),
    ("dec rewrite 384->768 3x3 8x256", 64, 384, (8, 256), (3, 3), (1, 1), (1, 1), True),
    ("time rewrite 48->96 k3 65536", 64, 48, (1, 65536), (1, 3), (0, 1), (1, 1), True),
    ("time rewrite 96->192 k3 16384", 64, 96, (1, 16384), (1, 3), (0, 1), (1, 1), True),
    ("time rewrite 192->384 k3 4096", 64, 192, (1, 4096), (1, 3), (0, 1), (1, 1), True),
    ("dconv conv3 48->12 d1 65536", 64, 48, (1, 65536), (1, 3), (0, 1), (1, 1), False),
    ("dconv conv3 96->24 d2 16384", 64, 96, (1, 16384), (1, 3), (0, 2), (1, 2), False),
    ("deep rewrite 768->1536 3x3 1x256", 64, 768, (1, 256), (3, 3), (1, 1), (1, 1), True),
    ("deep rewrite 1536->3072 k3 128", 64, 1536, (1, 128), (1, 3), (0, 1), (1, 1), True),
]


def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


if os.environ.get("PERF_LAYERS"):
    LAYERS = [LAYERS[int(i)] for i in os.environ["PERF_LAYERS"].split(",")]
for name, N, Cc, (A, B), ks, pad, dil, glu in LAYERS:
    M = 2 * Cc if glu else Cc // 4
    x = torch.randn(N, Cc, A, B, device=dev)
    w = torch.randn(M, Cc, *ks, device=dev) / (Cc * ks[0] * ks[1]) ** 0.5
    b = torch.zeros(M, device=dev)
    flops = 2.0 * N * A * B * M * Cc * ks[0] * ks[1]
    row = []
    for halo in (False, True):
        convplan.HALO = halo
        ops._PLANS.clear()
        xr = x.clone().requires_grad_(True)
        if glu:
            f = lambda: ops.conv2d_glu(xr, w, b, (1, 1), pad, dil)
        else:
            f = lambda: ops.conv2d(xr, w, b, (1, 1), pad, dil, out_bf16=True)
        t_f = timed(f)
        y = f()
        gy = torch.randn_like(y, dtype=torch.float32) if y.dtype == torch.float32 else torch.randn(y.shape, device=dev).to(y.dtype)
        t_fb = t_f if os.environ.get("PERF_FWD_ONLY") else timed(lambda: torch.autograd.grad(f(), xr, gy))
        row += [t_f, t_fb - t_f]
    print(f"{name:36s} fwd {row[0]:6.3f} -> {row[2]:6.3f} ms ({flops / row[2] / 1e9:5.0f} TF/s)   bwd(x) {row[1]:6.3f} -> {row[3]:6.3f} ms", flush=True)
