"""Probe for DESIGN 8.8: the forward gather-GEMM with its operand in bf16 CHANNELS-LAST layout ([N][A][B][C], one 16-byte load
per lane and K step: gemm_tap_kernel<R, 2, 3>) against the product path (fp32 [N][C][A][B], 8 loads + 4 converts per K step) on
the 3x3 / 1x1 layer shapes of the Demucs step.  Same weights, same (bf16-representable) input values: the results must agree
bit for bit; the timings say what the layout change buys on the B-operand path alone.   python scripts/probes/cl_gather_probe.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from remfx_amd import _lib, convplan, ops  # noqa: E402
from remfx_amd._lib import GemmDesc  # noqa: E402
from remfx_amd.ops import _ptr, _stream  # noqa: E402

ops.set_gemm_precision("bf16")
dev = torch.device("cuda:0")
SHAPES = [  # N, Cin, A, B, Cout, k, pad
    (64, 192, 32, 256, 384, 3, 1), (64, 384, 8, 256, 768, 3, 1), (64, 96, 128, 256, 192, 3, 1), (64, 48, 512, 256, 96, 3, 1),
    (64, 768, 8, 256, 384, 3, 1), (64, 96, 128, 256, 192, 1, 0), (64, 48, 512, 256, 96, 1, 0), (64, 384, 1, 1024, 384, 1, 0)]


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for N, Cin, A, B, Cout, k, pad in SHAPES:
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(N, Cin, A, B, generator=g) * 0.5).bfloat16().float().to(dev)          # bf16-representable values
    w = (torch.randn(Cout, Cin, k, k if A > 1 else k, generator=g) * 0.05).to(dev) if A > 1 else \
        (torch.randn(Cout, Cin, 1, k, generator=g) * 0.05).to(dev)
    kk = (k, k) if A > 1 else (1, k)
    pp = (pad, pad) if A > 1 else (0, pad)
    y_ref = torch.empty((N, Cout, A, B), device=dev)
    plan = convplan.conv_fwd_plan(tuple(x.shape), x.stride(), tuple(w.shape), (1, 1), pp, (1, 1), y_ref.stride())
    dp = ops.DevPlan(plan, dev)
    apk = ops.pack_a(dp, w.contiguous())
    ref = lambda: ops.gemm_fwd(dp, apk, x, y_ref)
    # channels-last bf16 operand viewed as (N, C, A, B)
    xcl = x.permute(0, 2, 3, 1).contiguous().bfloat16()                                     # [N][A][B][C]
    xv = xcl.permute(0, 3, 1, 2)                                                            # strides (A*B*C, 1, B*C, C)
    y_cl = torch.empty_like(y_ref)
    plan2 = convplan.conv_fwd_plan(tuple(xv.shape), xv.stride(), tuple(w.shape), (1, 1), pp, (1, 1), y_cl.stride())
    dp2 = ops.DevPlan(plan2, dev)
    apk2 = ops.pack_a(dp2, w.contiguous())
    d3 = GemmDesc.from_buffer_copy(dp2.desc)
    d3.in_bf16 = 3
    d3.in_extent = int(plan2.in_extent) * 2
    e = _lib.Epilogue()

    def cl():
        rc = _lib.lib().rfx_gemm_fwd(C.byref(d3), _ptr(apk2), _ptr(dp2.tap_tab), _ptr(xcl), _ptr(y_cl), C.byref(e), None, None, 0, 0,
                                     None, 2, _stream())
        assert rc == 0, rc
    t_ref, t_cl = timed(ref), timed(cl)
    flops = 2.0 * Cout * Cin * kk[0] * kk[1] * N * A * B
    same = torch.equal(y_ref, y_cl)
    err = float((y_ref - y_cl).abs().max())
    print(f"N={N} Cin={Cin} {A}x{B} Cout={Cout} k={kk} R={plan.R}: product {t_ref:.3f} ms ({flops / t_ref / 1e9:.0f} TF/s)  channels-last bf16 "
          f"{t_cl:.3f} ms ({flops / t_cl / 1e9:.0f} TF/s)  x{t_ref / t_cl:.2f}  bit-equal={same} max|diff|={err:.2e}")
