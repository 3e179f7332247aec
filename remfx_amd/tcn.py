"""micro-TCN on the gather-GEMM MFMA kernels (mirror of reference remfx/tcn.py).

Same constructor arguments, attribute names and state_dict keys as
``remfx.tcn.TCN`` (tcn.py:62-138): process_blocks.{n}.{conv1.weight, conv1.bias,
res.weight, relu.weight}, output.{weight, bias}.  nn.Conv1d / nn.PReLU objects are
kept only as parameter containers (so default init and checkpoints line up); the
arithmetic is one fused HIP launch per block:

    y = PReLU(conv1(x) + b) + crop(res(x))          tcn.py:48-59

computed as a two-phase gather-GEMM (phase 1: K = Cin*k dilated taps, PReLU applied
to the accumulators in registers, phase 2: K = Cin residual 1x1 at the crop offset).
Backward re-materialises the pre-activation instead of storing it (20 blocks x
256 x 262144 fp32 would not fit 288 GB at batch 32).
"""
from typing import Callable

import numpy as np
import torch
import torch.nn as nn

from . import convplan, ops
from .utils import causal_crop, center_crop, crop_start


def _shift_plan(xshape, xstrides, M, shift, gstrides=None, oshape=None, ostrides=None):
    """1x1 conv reading x[..., b + shift]: ktab rows (ci*cs + shift*bs, 0, shift)."""
    N, Cin, IA, IB = xshape
    ns, cs, as_, bs = xstrides
    ci = np.arange(Cin)
    ktab = np.stack([ci * cs + shift * bs, np.zeros_like(ci), np.full_like(ci, shift), np.zeros_like(ci)], -1)
    on, oc, oa, ob = ostrides
    p = convplan.GemmPlan(N=N, M=M, K=Cin, OA=oshape[2], OB=oshape[3], IA=IA, IB=IB, SA=1, SB=1,
                          in_ns=ns, in_as=as_, in_bs=bs, out_ns=on, out_cs=oc, out_as=oa, out_bs=ob,
                          ktab=ktab, woff=ci.copy(), w_ms=Cin, cin=Cin, in_cs=cs)
    return p.finalize()


def _block_plans(x4, Cout, ksize, dilation, causal):
    """Forward plans for one block on a (N, Cin, 1, L) input."""
    N, Cin, _, L = x4.shape
    Lout = L - (ksize - 1) * dilation
    if Lout <= 0:
        raise ValueError(f"input length {L} shorter than the block's receptive field")
    out_strides = (Cout * Lout, Lout, Lout, 1)
    key = ops._key("tcnblk", x4.shape, x4.stride(), Cout, ksize, dilation, causal)
    start = crop_start(causal, L, Lout)

    def build():
        p1 = convplan.conv_fwd_plan(tuple(x4.shape), x4.stride(), (Cout, Cin, 1, ksize), (1, 1), (0, 0),
                                    (1, dilation), out_strides)
        p2 = _shift_plan(tuple(x4.shape), x4.stride(), Cout, start, oshape=(N, Cout, 1, Lout),
                         ostrides=out_strides)
        p2.R, p2.Mpad = p1.R, p1.Mpad          # both phases share one launch geometry
        return [p1, p2]
    return ops._plans(key, x4.device, build), Lout, start


def tcn_block_forward(x, w1, b1, slope, wres, dilation, causal):
    """x: (N, Cin, L) -> (N, Cout, L - (k-1)*d); one fused launch."""
    ops._req(x, "x")
    x4 = x.unsqueeze(2)
    Cout, Cin, ksize = w1.shape
    (dp1, dp2), Lout, _ = _block_plans(x4, Cout, ksize, dilation, causal)
    out = torch.empty((x.shape[0], Cout, Lout), device=x.device, dtype=torch.float32)
    a1 = ops.pack_cached(dp1, w1)
    a2 = ops.pack_cached(dp2, wres)
    ops.gemm_fwd(dp1, a1, x4, out, bias=b1, act="prelu", act_param=slope, dp2=dp2, apack2=a2)
    return out


class TCNBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, slope, wres, dilation, causal):
        ctx.save_for_backward(x, w1, b1, slope, wres)
        ctx.cfg = (dilation, causal)
        return tcn_block_forward(x, w1, b1, slope, wres, dilation, causal)

    @staticmethod
    def backward(ctx, g):
        x, w1, b1, slope, wres = ctx.saved_tensors
        dilation, causal = ctx.cfg
        g = g.contiguous()
        N, Cin, L = x.shape
        Cout, _, ksize = w1.shape
        x4, g4 = x.unsqueeze(2), g.unsqueeze(2)
        (dp1, dp2), Lout, start = _block_plans(x4, Cout, ksize, dilation, causal)
        # 1. re-materialise pre = conv1(x)+b; g1 = g * prelu'(pre); dslope = sum g * min(pre, 0)
        g1 = torch.empty_like(g)
        dslope = torch.zeros((64, slope.numel()), device=x.device, dtype=torch.float32)   # partial sums, see gemm_fwd.h
        ops.gemm_fwd(dp1, ops.pack_cached(dp1, w1), x4, g1, bias=b1, act="prelu", act_param=slope,
                     res=g4, bwd=True, gparam=dslope)
        g14 = g1.unsqueeze(2)
        # 2. weight gradients: conv1 from (x, g1) with the bias row; residual 1x1 from (x shifted, g)
        dw1, db1 = ops.conv2d_wgrad(x4, g14, (Cout, Cin, 1, ksize), (1, 1), (0, 0), (1, dilation), True)
        p2 = dp2.p
        dap = ops.gemm_wgrad(dp2, x4, g4)
        dwres = torch.zeros_like(wres)
        ops.unpack_add(dp2, dap, dwres)
        # 3. input gradient: conv1^T over g1 (phase 1) + res^T over g at the crop offset (phase 2)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            dx4 = dx.unsqueeze(2)
            key = ops._key("tcnblk_dx", x4.shape, x4.stride(), Cout, ksize, dilation, causal)

            def build():
                pd = convplan.conv_dgrad_plans(tuple(x4.shape), x4.stride(), (Cout, Cin, 1, ksize), (1, 1),
                                               (0, 0), (1, dilation), tuple(g4.shape), g4.stride())
                assert len(pd) == 1
                # residual transpose: dx[ci, i] += sum_co wres[co, ci] g[co, i - start]
                co = np.arange(Cout)
                kt = np.stack([co * g4.stride(1) - start * g4.stride(3), np.zeros_like(co),
                               np.full_like(co, -start), np.zeros_like(co)], -1)
                pr = convplan.GemmPlan(N=N, M=Cin, K=Cout, OA=1, OB=L, IA=1, IB=Lout, SA=1, SB=1,
                                       in_ns=g4.stride(0), in_as=g4.stride(2), in_bs=g4.stride(3),
                                       out_ns=x4.stride(0), out_cs=x4.stride(1), out_as=x4.stride(2),
                                       out_bs=x4.stride(3), ktab=kt, woff=co * Cin, w_ms=1, cin=Cout,
                                       in_cs=g4.stride(1)).finalize()
                pr.R, pr.Mpad = pd[0].R, pd[0].Mpad
                return [pd[0], pr]
            dpd, dpr = ops._plans(key, x.device, build)
            ops.gemm_fwd(dpd, ops.pack_cached(dpd, w1), g14, dx4, dp2=dpr,
                         apack2=ops.pack_cached(dpr, wres), in2=g4)
        return dx, dw1.view_as(w1), db1, dslope.sum(0).view_as(slope), dwres, None, None


class TCNBlock(nn.Module):
    """Parameter container + fused forward (reference tcn.py:11-59)."""

    def __init__(self, in_ch: int, out_ch: int, kernel_size: int = 3, dilation: int = 1, stride: int = 1,
                 crop_fn: Callable = causal_crop) -> None:
        super().__init__()
        if stride != 1:
            raise ValueError("TCNBlock: only stride 1 is used by the reference (tcn.py:115)")
        self.in_ch, self.out_ch, self.kernel_size, self.stride = in_ch, out_ch, kernel_size, stride
        self.dilation = dilation
        self.crop_fn = crop_fn
        self.conv1 = nn.Conv1d(in_ch, out_ch, kernel_size, stride=stride, padding=0, dilation=dilation, bias=True)
        self.res = nn.Conv1d(in_ch, out_ch, kernel_size=1, groups=1, stride=stride, bias=False)
        self.relu = nn.PReLU(out_ch)

    def forward(self, x):
        return TCNBlockFn.apply(x, self.conv1.weight, self.conv1.bias, self.relu.weight, self.res.weight,
                                self.dilation, self.crop_fn is causal_crop)


class TCN(nn.Module):
    def __init__(self, ninputs: int = 1, noutputs: int = 1, nblocks: int = 4, channel_growth: int = 0,
                 channel_width: int = 32, kernel_size: int = 13, stack_size: int = 10,
                 dilation_growth: int = 10, condition: bool = False, latent_dim: int = 2,
                 norm_type: str = "identity", causal: bool = False, estimate_loudness: bool = False) -> None:
        super().__init__()
        self.ninputs, self.noutputs, self.nblocks = ninputs, noutputs, nblocks
        self.channel_growth, self.channel_width, self.kernel_size = channel_growth, channel_width, kernel_size
        self.stack_size, self.dilation_growth = stack_size, dilation_growth
        self.condition, self.latent_dim, self.norm_type = condition, latent_dim, norm_type
        self.causal, self.estimate_loudness = causal, estimate_loudness
        self.crop_fn = causal_crop if causal else center_crop          # tcn.py:94-97
        if estimate_loudness:
            self.loudness = nn.Linear(latent_dim, 1)
        self.process_blocks = nn.ModuleList()
        out_ch = -1
        for n in range(nblocks):
            in_ch = out_ch if n > 0 else ninputs
            out_ch = in_ch * channel_growth if channel_growth > 1 else channel_width
            self.process_blocks.append(TCNBlock(in_ch, out_ch, kernel_size,
                                                dilation_growth ** (n % stack_size), stride=1,
                                                crop_fn=self.crop_fn))
        self.output = nn.Conv1d(out_ch, noutputs, kernel_size=1)
        self.receptive_field = self.compute_receptive_field()
        self.block_size = 2048
        self.buffer = torch.zeros(2, self.receptive_field + self.block_size - 1)   # plain tensor, as upstream

    def forward(self, x):
        ops._req(x, "x")
        x = x.contiguous()
        for block in self.process_blocks:
            x = block(x)
        y = ops.conv1d(x, self.output.weight, self.output.bias)
        return ops.activation(y, "tanh")                                  # tcn.py:129

    def compute_receptive_field(self):
        rf = self.kernel_size
        for n in range(1, self.nblocks):
            rf += (self.kernel_size - 1) * self.dilation_growth ** (n % self.stack_size)
        return rf
