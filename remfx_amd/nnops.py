"""Layer-level ops used by the removal networks.

Every function here dispatches to a hand-written HIP kernel (csrc/*.hip) through the C
ABI.  Ops whose kernel is not written yet run through torch-ROCm on the GPU and are
recorded in ``INTERIM`` -- they never fall back to the CPU or to the oracle, and
``RFX_STRICT_NATIVE=1`` turns any interim op into an error (DESIGN.md keeps the
coverage table).
"""
import os

import torch
import torch.nn.functional as F

from . import _lib, ops
from ._lib import check
from .ops import _ptr, _stream

INTERIM = set()


def _interim(name):
    if os.environ.get("RFX_STRICT_NATIVE", "0") == "1":
        raise RuntimeError(f"op '{name}' has no HIP kernel yet (RFX_STRICT_NATIVE=1)")
    INTERIM.add(name)


def gelu(x):
    return ops.activation(x, "gelu")


GN_MODES = {"none": 0, "gelu": 1, "glu": 2, "glu_scale_res": 3}


class _GroupNormFn(torch.autograd.Function):
    """GroupNorm fused with GELU / GLU / (res + scale * GLU); backward re-materialises gn(x)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, mode, res, scale, sums=None):
        x16 = x.dtype == torch.bfloat16            # a conv output kept in 16 bits (bf16 mode, ops.bf16_storage)
        if not x16:
            ops._req(x, "x")
        x = x.contiguous()
        N, Cc = x.shape[0], x.shape[1]
        S = x.numel() // (N * Cc)
        oshape = list(x.shape)
        if mode >= 2:
            oshape[1] = Cc // 2
        y = torch.empty(oshape, device=x.device, dtype=torch.float32)
        mean = torch.empty(N * groups, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        given = (sums.shape[1] if sums.dim() == 3 else 1) if sums is not None else 0     # number of partial-sum slots
        if sums is None:
            # the statistics kernel stores one pair per (group, chunk) into its own slot: no zero fill, no atomics
            chunks = int(_lib.lib().rfx_groupnorm_stat_chunks(Cc, S, groups))
            sums = torch.empty(N * groups * 2 * max(chunks, 1), device=x.device, dtype=torch.float64)
            given = -1
        if res is not None:
            res = res.contiguous()
        fwd = _lib.lib().rfx_groupnorm_fwd_x16 if x16 else _lib.lib().rfx_groupnorm_fwd
        check(fwd(_ptr(x), _ptr(gamma), _ptr(beta), N, Cc, S, groups, eps, mode,
                  _ptr(res), _ptr(scale), _ptr(sums), given, _ptr(mean), _ptr(rstd), _ptr(y), _stream()),
              "rfx_groupnorm_fwd")
        ctx.save_for_backward(x, gamma, beta, mean, rstd, scale)
        ctx.cfg = (N, Cc, S, groups, mode)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, gamma, beta, mean, rstd, scale = ctx.saved_tensors
        N, Cc, S, groups, mode = ctx.cfg
        gy = gy.contiguous()
        dx = torch.empty_like(x)
        # parameter gradients: straight into the flat gradient buffer when a GradSink is armed and this is the parameter's first
        # gradient of the step (the kernel SETS its outputs); otherwise fresh tensors for autograd's accumulation
        sink = ops.SINK
        tg = tb = ts = None
        if sink is not None:
            tg, tb = sink.lookup(gamma), sink.lookup(beta)
            ts = sink.lookup(scale) if mode == 3 else None
            ok = (tg is not None and tb is not None and (mode != 3 or ts is not None) and
                  all(sink.writes[t[0]] == 0 for t in (tg, tb, ts) if t is not None))
            if not ok:
                tg = tb = ts = None
        dgamma = tg[1] if tg else torch.empty_like(gamma)
        dbeta = tb[1] if tb else torch.empty_like(beta)
        dscale = (ts[1] if ts else torch.empty_like(scale)) if mode == 3 else None
        gsum = torch.empty(int(_lib.lib().rfx_norm_bwd_work_floats(N, Cc, S, groups)), device=x.device, dtype=torch.float32)
        bwd = _lib.lib().rfx_groupnorm_bwd_x16 if x.dtype == torch.bfloat16 else _lib.lib().rfx_groupnorm_bwd
        check(bwd(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(rstd), _ptr(gy),
                  N, Cc, S, groups, mode, _ptr(scale), _ptr(gsum), _ptr(dx),
                  _ptr(dgamma), _ptr(dbeta), _ptr(dscale), _stream()),
              "rfx_groupnorm_bwd")
        if tg:
            for t in (tg, tb, ts):
                if t is not None:
                    sink.wrote(t[0])
            return dx, None, None, None, None, None, (gy if mode == 3 else None), None, None
        return dx, dgamma, dbeta, None, None, None, (gy if mode == 3 else None), dscale, None


def group_norm(x, groups, weight, bias, eps=1e-5, mode="none", res=None, scale=None, sums=None):
    """mode: none | gelu | glu | glu_scale_res (out = res + scale[c] * glu(gn(x))).
    sums: fp64 (N*groups, 2) or (N*groups, slots, 2) {sum, sum^2} already accumulated by the producing GEMM's epilogue."""
    return _GroupNormFn.apply(x, weight, bias, groups, eps, GN_MODES[mode], res, scale, sums)


class _BatchNormFn(torch.autograd.Function):
    """BatchNorm2d (+ReLU).  training=True: batch statistics (and running-stat update by the caller
    from the returned mean / biased var); training=False: running statistics, forward only."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, relu):
        ops._req(x, "x")
        x = x.contiguous()
        N, Cc = x.shape[0], x.shape[1]
        S = x.numel() // (N * Cc)
        y = torch.empty_like(x)
        mode = 4 if relu else 0
        if training:
            mean = torch.empty(Cc, device=x.device, dtype=torch.float32)
            rstd = torch.empty_like(mean)
            sums = torch.empty(Cc * 2 * int(_lib.lib().rfx_batchnorm_stat_slots(N, S)), device=x.device, dtype=torch.float64)
            given = 0
        else:
            mean = running_mean.contiguous()
            rstd = torch.rsqrt(running_var + eps)
            sums, given = None, 1
        check(_lib.lib().rfx_batchnorm_fwd(_ptr(x), _ptr(gamma), _ptr(beta), N, Cc, S, eps, mode, given,
                                           _ptr(sums), _ptr(mean), _ptr(rstd), _ptr(y), _stream()),
              "rfx_batchnorm_fwd")
        if training and running_mean is not None:
            with torch.no_grad():              # C-length vectors: bookkeeping, nn.BatchNorm2d semantics
                n = N * S
                var = 1.0 / (rstd * rstd) - eps
                running_mean.mul_(1 - momentum).add_(mean, alpha=momentum)
                running_var.mul_(1 - momentum).add_(var * (n / max(n - 1, 1)), alpha=momentum)
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        ctx.cfg = (N, Cc, S, mode, training)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        N, Cc, S, mode, training = ctx.cfg
        if not training:
            raise RuntimeError("batch_norm backward in eval mode is not on the reference's path")
        gy = gy.contiguous()
        dx = torch.empty_like(x)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
        work = torch.empty(int(_lib.lib().rfx_norm_bwd_work_floats(N, Cc, S, 0)), device=x.device, dtype=torch.float32)
        check(_lib.lib().rfx_batchnorm_bwd(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(rstd), _ptr(gy),
                                           N, Cc, S, mode, _ptr(work), _ptr(dx), _ptr(dgamma), _ptr(dbeta),
                                           _stream()), "rfx_batchnorm_bwd")
        return dx, dgamma, dbeta, None, None, None, None, None, None


def batch_norm(x, bn, training, relu=False):
    """bn: nn.BatchNorm2d parameter container (weight, bias, running_mean, running_var)."""
    if training and bn.num_batches_tracked is not None:
        bn.num_batches_tracked += 1
    return _BatchNormFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, bn.momentum,
                              bn.eps, relu)


class _AvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kh, kw):
        ops._req(x, "x")
        x = x.contiguous()
        N, Cc, H, W = x.shape
        y = torch.empty((N, Cc, H // kh, W // kw), device=x.device, dtype=torch.float32)
        check(_lib.lib().rfx_avgpool2d_fwd(_ptr(x), _ptr(y), N * Cc, H, W, kh, kw, _stream()), "rfx_avgpool2d_fwd")
        ctx.cfg = (N, Cc, H, W, kh, kw)
        return y

    @staticmethod
    def backward(ctx, gy):
        N, Cc, H, W, kh, kw = ctx.cfg
        gx = torch.empty((N, Cc, H, W), device=gy.device, dtype=torch.float32)
        check(_lib.lib().rfx_avgpool2d_bwd(_ptr(gy.contiguous()), _ptr(gx), N * Cc, H, W, kh, kw, _stream()),
              "rfx_avgpool2d_bwd")
        return gx, None, None


def avg_pool2d(x, kernel_size):
    kh, kw = kernel_size
    if (kh, kw) == (1, 1):
        return x
    return _AvgPoolFn.apply(x, kh, kw)


class _GluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ops._req(x, "x")
        x = x.contiguous()
        N, Cc = x.shape[0], x.shape[1]
        S = x.numel() // (N * Cc)
        shp = list(x.shape)
        shp[1] = Cc // 2
        y = torch.empty(shp, device=x.device, dtype=torch.float32)
        check(_lib.lib().rfx_glu_fwd(_ptr(x), _ptr(y), N, Cc, S, _stream()), "rfx_glu_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        N, Cc = x.shape[0], x.shape[1]
        S = x.numel() // (N * Cc)
        gx = torch.empty_like(x)
        check(_lib.lib().rfx_glu_bwd(_ptr(x), _ptr(gy.contiguous()), _ptr(gx), N, Cc, S, _stream()), "rfx_glu_bwd")
        return gx


def glu(x, dim=1):
    if dim != 1:
        raise ValueError("glu: channel axis (dim=1) only")
    return _GluFn.apply(x)


class _AddFn(torch.autograd.Function):
    """x + alpha * y with y broadcast over any of the 4 dims (N, C, A, B); x contiguous."""

    @staticmethod
    def forward(ctx, x, y, alpha):
        ops._req(x, "x")
        x = x.contiguous()
        x4 = x if x.dim() == 4 else x.unsqueeze(2)
        y4 = y if y.dim() == 4 else y.unsqueeze(2)
        ye = y4.expand_as(x4)
        out = torch.empty_like(x4)
        N, Cc, A, B = x4.shape
        ys = ye.stride()
        check(_lib.lib().rfx_add_bcast(_ptr(x4), _ptr(ye), _ptr(out), N, Cc, A, B, ys[0], ys[1], ys[2], ys[3],
                                       float(alpha), _stream()), "rfx_add_bcast")
        ctx.cfg = (tuple(y4.shape), tuple(x4.shape), alpha, y.dim())
        return out if x.dim() == 4 else out.squeeze(2)

    @staticmethod
    def backward(ctx, g):
        yshape, xshape, alpha, ydim = ctx.cfg
        gy = None
        if ctx.needs_input_grad[1]:
            g4 = g if g.dim() == 4 else g.unsqueeze(2)
            if yshape == xshape:
                gy = g4 * alpha if alpha != 1.0 else g4
            elif yshape[0] == 1 and yshape[3] == 1 and yshape[1:3] == xshape[1:3]:
                # reduce over (N, B): (c, a) plays the channel role of rfx_channel_sum
                N, Cc, A, B = xshape
                gy = ops.channel_sum(g4.contiguous().view(N, Cc * A, 1, B)).view(1, Cc, A, 1) * alpha
            else:
                gy = g4.sum_to_size(yshape) * alpha
            if ydim == 3:
                gy = gy.squeeze(2)
        return g, gy, None


def add(x, y, alpha=1.0):
    return _AddFn.apply(x, y, alpha)


class _DConvLayerFn(torch.autograd.Function):
    """One depth-layer of the Hybrid Demucs DConv branch, ONE launch per direction (csrc/dconv.hip):
    x + scale * GLU(GN(conv1x1(GELU(GN(conv3_dilated(x)))))) on (N, 48, 256) samples, bf16 arithmetic.
    DCONV_FUSED_BWD (round 4): only the layer INPUT is saved; the backward launch recomputes the forward, returns dL/dx and the
    LayerScale / GroupNorm gradients (per-workgroup partial rows, summed here) and hands dz / a / dh to the two weight-gradient GEMMs.
    Otherwise (round 3 path, bench.py --no-fused-dconv-bwd): the forward also stores h / z (bf16), a and the GroupNorm statistics and
    the backward runs layer by layer on the existing kernels (GroupNorm backward x2, input-gradient GEMM x2, weight-gradient GEMM x2)."""

    @staticmethod
    def forward(ctx, x, w1, b1, g1w, g1b, w2, b2, g2w, g2b, scale, dil, eps, grad_on=True):
        ops._req(x, "x")
        N, Cc, T = x.shape
        H = Cc // 4
        out = torch.empty_like(x)
        # grad_on: the CALLER's grad mode (inside Function.forward it is always off, and needs_input_grad ignores no_grad)
        need = grad_on and any(ctx.needs_input_grad)
        fused_bwd = need and DCONV_FUSED_BWD
        h16 = z16 = a_out = stats = None
        if need and not fused_bwd:
            h16 = torch.empty((N, H, T), device=x.device, dtype=torch.bfloat16)
            z16 = torch.empty((N, 2 * Cc, T), device=x.device, dtype=torch.bfloat16)
            a_out = torch.empty((N, H, T), device=x.device, dtype=torch.float32)
            stats = torch.empty((4, N), device=x.device, dtype=torch.float32)
        check(_lib.lib().rfx_dconv_layer_fwd(_ptr(x), _ptr(out), N, Cc, T, dil, _ptr(w1), _ptr(b1), _ptr(g1w), _ptr(g1b),
                                             _ptr(w2), _ptr(b2), _ptr(g2w), _ptr(g2b), _ptr(scale), float(eps), _ptr(h16),
                                             _ptr(z16), _ptr(a_out), _ptr(stats), _stream()), "rfx_dconv_layer_fwd")
        if fused_bwd:
            ctx.save_for_backward(x, w1, b1, g1w, g1b, w2, b2, g2w, g2b, scale)
            ctx.cfg = (dil, float(eps), True)
        elif need:
            ctx.save_for_backward(x, w1, b1, g1w, g1b, w2, b2, g2w, g2b, scale, h16, z16, a_out, stats)
            ctx.cfg = (dil, float(eps), False)
        return out

    @staticmethod
    def backward(ctx, gy):
        dil, eps, fused_bwd = ctx.cfg
        if fused_bwd:
            return _DConvLayerFn._backward_fused(ctx, gy, dil, eps)
        x, w1, b1, g1w, g1b, w2, b2, g2w, g2b, scale, h16, z16, a_out, stats = ctx.saved_tensors
        N, Cc, T = x.shape
        H = Cc // 4
        L = _lib.lib()
        gy = gy.contiguous()
        # GroupNorm(1, 2C) + GLU + LayerScale backward (mode 3): dz (bf16), dgamma2 / dbeta2 / dscale; the residual passes gy on
        dz = torch.empty_like(z16)
        dg2w, dg2b, dscale = torch.empty_like(g2w), torch.empty_like(g2b), torch.empty_like(scale)
        work = torch.empty(int(L.rfx_norm_bwd_work_floats(N, 2 * Cc, T, 1)), device=x.device, dtype=torch.float32)
        check(L.rfx_groupnorm_bwd_x16(_ptr(z16), _ptr(g2w), _ptr(g2b), _ptr(stats[2]), _ptr(stats[3]), _ptr(gy), N, 2 * Cc, T, 1, 3,
                                      _ptr(scale), _ptr(work), _ptr(dz), _ptr(dg2w), _ptr(dg2b), _ptr(dscale), _stream()),
              "rfx_groupnorm_bwd")
        # 1x1 convolution: da = W2^T dz;  dW2, db2
        w24 = w2.reshape(2 * Cc, H, 1, 1)
        a4, dz4 = a_out.unsqueeze(2), dz.unsqueeze(2)
        da = ops.conv2d_dgrad(dz4, w24, tuple(a4.shape), tuple(a4.stride()), (1, 1), (0, 0), (1, 1)).squeeze(2)
        dw2, db2 = ops.conv2d_wgrad(a4, dz4, (2 * Cc, H, 1, 1), (1, 1), (0, 0), (1, 1), True, w2, b2)
        # GroupNorm(1, H) + GELU backward (mode 1): dh (bf16), dgamma1 / dbeta1
        dh = torch.empty_like(h16)
        dg1w, dg1b = torch.empty_like(g1w), torch.empty_like(g1b)
        work1 = torch.empty(int(L.rfx_norm_bwd_work_floats(N, H, T, 1)), device=x.device, dtype=torch.float32)
        check(L.rfx_groupnorm_bwd_x16(_ptr(h16), _ptr(g1w), _ptr(g1b), _ptr(stats[0]), _ptr(stats[1]), _ptr(da), N, H, T, 1, 1,
                                      None, _ptr(work1), _ptr(dh), _ptr(dg1w), _ptr(dg1b), None, _stream()), "rfx_groupnorm_bwd")
        # dilated convolution: dx = W1^T * dh + gy (the residual rides in the GEMM's store);  dW1, db1
        w14 = w1.reshape(H, Cc, 1, 3)
        x4, dh4 = x.unsqueeze(2), dh.unsqueeze(2)
        dx = ops.conv2d_dgrad(dh4, w14, tuple(x4.shape), tuple(x4.stride()), (1, 1), (0, dil), (1, dil), res=gy.unsqueeze(2)).squeeze(2)
        dw1, db1 = ops.conv2d_wgrad(x4, dh4, (H, Cc, 1, 3), (1, 1), (0, dil), (1, dil), True, w1, b1)
        dw2 = dw2.view_as(w2) if dw2 is not None else None           # each convolution takes the sink route (None) or not on its own
        dw1 = dw1.view_as(w1) if dw1 is not None else None
        return dx, dw1, db1, dg1w, dg1b, dw2, db2, dg2w, dg2b, dscale, None, None, None

    @staticmethod
    def _backward_fused(ctx, gy, dil, eps):
        x, w1, b1, g1w, g1b, w2, b2, g2w, g2b, scale = ctx.saved_tensors
        N, Cc, T = x.shape
        H = Cc // 4
        L = _lib.lib()
        gy = gy.contiguous()
        dx = torch.empty_like(x)
        dz = torch.empty((N, 2 * Cc, T), device=x.device, dtype=torch.bfloat16)
        a_out = torch.empty((N, H, T), device=x.device, dtype=torch.float32)
        dh = torch.empty((N, H, T), device=x.device, dtype=torch.bfloat16)
        partial = torch.empty((L.rfx_dconv_layer_bwd_rows(N), 5 * Cc + 2 * H), device=x.device, dtype=torch.float32)
        check(L.rfx_dconv_layer_bwd(_ptr(x), _ptr(gy), _ptr(dx), N, Cc, T, dil, _ptr(w1), _ptr(b1), _ptr(g1w), _ptr(g1b), _ptr(w2),
                                    _ptr(b2), _ptr(g2w), _ptr(g2b), _ptr(scale), eps, _ptr(dz), _ptr(a_out), _ptr(dh), _ptr(partial),
                                    _stream()), "rfx_dconv_layer_bwd")
        ps = partial.sum(0)
        dscale, dg2w, dg2b = ps[:Cc], ps[Cc:3 * Cc], ps[3 * Cc:5 * Cc]
        dg1w, dg1b = ps[5 * Cc:5 * Cc + H], ps[5 * Cc + H:]
        # weight / bias gradients of the two convolutions on the existing wgrad kernels (side stream + in place when a GradSink is armed)
        dw2, db2 = ops.conv2d_wgrad(a_out.unsqueeze(2), dz.unsqueeze(2), (2 * Cc, H, 1, 1), (1, 1), (0, 0), (1, 1), True, w2, b2)
        dw1, db1 = ops.conv2d_wgrad(x.unsqueeze(2), dh.unsqueeze(2), (H, Cc, 1, 3), (1, 1), (0, dil), (1, dil), True, w1, b1)
        dw2 = dw2.view_as(w2) if dw2 is not None else None
        dw1 = dw1.view_as(w1) if dw1 is not None else None
        return dx, dw1, db1, dg1w, dg1b, dw2, db2, dg2w, dg2b, dscale, None, None, None


# Round 4 measurement (N = 32768 x (48, 256), two depth layers, scripts/perf_dconv.py): fused forward without the h / z / a stores 2.5 ms
# (3.9 with them), but the fused backward launch takes 13.7 ms against 3.0 ms for the four kernels it replaces -- 7 ms of that in the
# LDS float atomics of the parameter-gradient row sums (~100 clocks per ds_add_f32 wave-instruction), the rest in a dependency-bound
# 16-wave lockstep with 424 bytes of scratch per lane.  Correct (tests/test_gpu_dconv_fused.py runs both), not the default.
DCONV_FUSED_BWD = False       # bench.py --fused-dconv-bwd switches the one-launch backward on
DCONV_FUSED = True            # bench.py --no-fused-dconv flips it for A/B runs


def dconv_layer_fused_ok(x, hidden, kernel_size, dil, need_grad):
    """Shapes / mode the fused DConv layer kernel takes (csrc/dconv.hip): bf16 arithmetic, contiguous fp32 (N, 48, 256),
    hidden = 12, kernel 3, dilation 1 or 2."""
    if not DCONV_FUSED or ops.GEMM_PREC != 2 or x.dim() != 3 or x.dtype != torch.float32 or not x.is_cuda or not x.is_contiguous():
        return False
    Cc, T = x.shape[1], x.shape[2]
    return hidden * 4 == Cc and kernel_size == 3 and bool(_lib.lib().rfx_dconv_layer_ok(Cc, T, dil))


def dconv_layer(x, conv1, gn1, conv2, gn2, scale, dil):
    """conv1 / conv2: nn.Conv1d parameter containers, gn1 / gn2: nn.GroupNorm(1, .), scale: the LayerScale vector."""
    return _DConvLayerFn.apply(x, conv1.weight, conv1.bias, gn1.weight, gn1.bias, conv2.weight, conv2.bias, gn2.weight, gn2.bias,
                               scale, int(dil), float(gn1.eps), torch.is_grad_enabled())


class _MulFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ops._req(a, "a"); ops._req(b, "b")
        a, b = a.contiguous(), b.contiguous()
        if a.shape != b.shape:
            raise ValueError("mul: equal shapes")
        y = torch.empty_like(a)
        check(_lib.lib().rfx_mul(_ptr(a), _ptr(b), _ptr(y), a.numel(), _stream()), "rfx_mul")
        ctx.save_for_backward(a, b)
        return y

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        ga, gb = torch.empty_like(a), torch.empty_like(b)
        check(_lib.lib().rfx_mul(_ptr(g), _ptr(b), _ptr(ga), a.numel(), _stream()), "rfx_mul")
        check(_lib.lib().rfx_mul(_ptr(g), _ptr(a), _ptr(gb), a.numel(), _stream()), "rfx_mul")
        return ga, gb


def mul(a, b):
    """a * b elementwise (equal shapes) with gradients to both."""
    return _MulFn.apply(a, b)


def row_standardize(x, eps):
    """(x - mean) / (eps + std) over all dims but the first, unbiased std (HDemucs forward); x carries no
    gradient (it is the input waveform / its STFT).  Returns y, mean (R,), std (R,)."""
    ops._req(x, "x")
    x = x.contiguous()
    R = x.shape[0]
    L = x.numel() // R
    lib = _lib.lib()
    sums = torch.empty(2 * R * lib.rfx_row_moments_slots(L), device=x.device, dtype=torch.float64)    # one slot per workgroup
    mean = torch.empty(R, device=x.device, dtype=torch.float32)
    std, a, b = torch.empty_like(mean), torch.empty_like(mean), torch.empty_like(mean)
    check(lib.rfx_row_moments(_ptr(x), R, L, _ptr(sums), _ptr(mean), _ptr(std), float(eps), _ptr(a), _ptr(b), _stream()),
          "rfx_row_moments")                                   # a = 1 / (eps + std), b = -mean a come out of the finalize kernel
    y = torch.empty_like(x)
    check(_lib.lib().rfx_row_affine(_ptr(x), _ptr(a), _ptr(b), _ptr(y), R, L, _stream()), "rfx_row_affine")
    return y, mean, std


class _RowAffineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, a, b):
        x = x.contiguous()
        R = x.shape[0]
        y = torch.empty_like(x)
        check(_lib.lib().rfx_row_affine(_ptr(x), _ptr(a), _ptr(b), _ptr(y), R, x.numel() // R, _stream()), "rfx_row_affine")
        ctx.save_for_backward(a)
        return y

    @staticmethod
    def backward(ctx, g):
        (a,) = ctx.saved_tensors
        g = g.contiguous()
        R = g.shape[0]
        gx = torch.empty_like(g)
        check(_lib.lib().rfx_row_affine(_ptr(g), _ptr(a), None, _ptr(gx), R, g.numel() // R, _stream()), "rfx_row_affine")
        return gx, None, None


def row_moments(x, eps):
    """Per-row (leading dim) mean, unbiased std and the standardisation coefficients a = 1 / (eps + std), b = -mean a of a tensor that
    carries no gradient -- row_standardize without the affine pass (the consumer applies x a + b itself: clchain.head_conv_fm)."""
    ops._req(x, "x")
    x = x.contiguous()
    R = x.shape[0]
    L = x.numel() // R
    lib = _lib.lib()
    sums = torch.empty(2 * R * lib.rfx_row_moments_slots(L), device=x.device, dtype=torch.float64)
    mean = torch.empty(R, device=x.device, dtype=torch.float32)
    std, a, b = torch.empty_like(mean), torch.empty_like(mean), torch.empty_like(mean)
    check(lib.rfx_row_moments(_ptr(x), R, L, _ptr(sums), _ptr(mean), _ptr(std), float(eps), _ptr(a), _ptr(b), _stream()), "rfx_row_moments")
    return mean, std, a, b


class _CmToFmAffineFn(torch.autograd.Function):
    """y[n][frame][bin][c] = x[n][c][bin][frame] a[n] + b[n]: HDemucs' de-standardisation fused with the layout change in front of the
    frame-major inverse STFT (rfx_fm_cm_affine); the gradient reaches x only (a, b are detached statistics of the input)."""

    @staticmethod
    def forward(ctx, x, a, b):
        ops._req(x, "x")
        x = x.contiguous()
        N, Cc, bins, F = x.shape
        if Cc != 2 or x.dtype != torch.float32:
            raise ValueError("cm_to_fm_affine: (N, 2, bins, frames) fp32")
        y = torch.empty((N, F, bins, 2), device=x.device, dtype=torch.float32)
        check(_lib.lib().rfx_fm_cm_affine(_ptr(x), _ptr(y), _ptr(a), _ptr(b), N, bins, F, 1, _stream()), "rfx_fm_cm_affine")
        ctx.save_for_backward(a)
        return y

    @staticmethod
    def backward(ctx, g):
        (a,) = ctx.saved_tensors
        g = g.contiguous()
        N, F, bins, _ = g.shape
        gx = torch.empty((N, 2, bins, F), device=g.device, dtype=torch.float32)
        check(_lib.lib().rfx_fm_cm_affine(_ptr(g), _ptr(gx), _ptr(a), None, N, bins, F, 0, _stream()), "rfx_fm_cm_affine")
        return gx, None, None


def cm_to_fm_affine(x, a, b):
    return _CmToFmAffineFn.apply(x, a.contiguous(), b.contiguous())


def row_affine(x, a, b):
    """x * a[r] + b[r] per leading-dim row (de-standardisation); gradient flows to x only."""
    return _RowAffineFn.apply(x, a.contiguous(), b.contiguous())


class _LocalStateFn(torch.autograd.Function):
    """softmax_t(k^T q / sqrt(ch) + decay penalty, masked diagonal) applied to the content: one HIP launch per direction
    (csrc/attention.hip) instead of two rocBLAS einsums, an ATen softmax and the (B, h, T, T) round trips."""

    @staticmethod
    def forward(ctx, q, k, cont, qd, heads, ndecay):
        for t, n in ((q, "q"), (k, "k"), (cont, "content"), (qd, "query_decay")):
            ops._req(t, n)
        q, k, cont, qd = q.contiguous(), k.contiguous(), cont.contiguous(), qd.contiguous()
        B, Ctot, T = q.shape
        ch = Ctot // heads
        out = torch.empty_like(q)
        need_w = any(ctx.needs_input_grad[:4])
        L = _lib.lib()
        # bf16 mode: MFMA kernels (attention_mfma.hip), weights recomputed in the backward pass instead of stored
        mfma = ops.GEMM_PREC == 2 and bool(L.rfx_localstate_mfma_ok(B, heads, ch, T, ndecay))
        # more than 256 frames (whole files) or ch * T beyond the LDS-resident kernel: the streaming any-T kernels
        gen = not mfma and (T > 256 or ch * T > 12288 or ndecay > 8)
        if mfma:
            check(L.rfx_localstate_mfma_fwd(_ptr(q), _ptr(k), _ptr(cont), _ptr(qd), B, heads, ch, T, ndecay, _ptr(out),
                                            _stream()), "rfx_localstate_mfma_fwd")
            if need_w:
                ctx.save_for_backward(q, k, cont, qd)
        elif gen:
            if ch > 104 or ndecay > 64:
                raise ValueError(f"LocalState attention: {ch} channels per head / {ndecay} decay terms exceed the HIP kernels' "
                                 "limits (ch <= 104, ndecay <= 64)")
            stat = torch.empty((B * heads * T, 4), device=q.device, dtype=torch.float32)
            check(L.rfx_localstate_gen_fwd(_ptr(q), _ptr(k), _ptr(cont), _ptr(qd), B, heads, ch, T, ndecay, _ptr(stat),
                                           _ptr(out), _stream()), "rfx_localstate_gen_fwd")
            if need_w:
                ctx.save_for_backward(q, k, cont, qd, stat, out)
        else:
            w = torch.empty((B, heads, T, T), device=q.device, dtype=torch.float32) if need_w else None
            check(L.rfx_localstate_fwd(_ptr(q), _ptr(k), _ptr(cont), _ptr(qd), B, heads, ch, T, ndecay, _ptr(w),
                                       _ptr(out), _stream()), "rfx_localstate_fwd")
            if need_w:
                ctx.save_for_backward(q, k, cont, qd, w)
        ctx.cfg = (B, heads, ch, T, ndecay, mfma, gen)
        return out

    @staticmethod
    def backward(ctx, g):
        B, heads, ch, T, ndecay, mfma, gen = ctx.cfg
        g = g.contiguous()
        q, k, cont, qd = ctx.saved_tensors[:4]
        dq, dk, dc, dqd = torch.empty_like(q), torch.empty_like(k), torch.empty_like(cont), torch.empty_like(qd)
        if mfma:
            stat = torch.empty((B * heads * T, 4), device=q.device, dtype=torch.float32)
            check(_lib.lib().rfx_localstate_mfma_bwd(_ptr(q), _ptr(k), _ptr(cont), _ptr(qd), _ptr(g), B, heads, ch, T, ndecay,
                                                     _ptr(dq), _ptr(dk), _ptr(dc), _ptr(dqd), _ptr(stat), _stream()),
                  "rfx_localstate_mfma_bwd")
        elif gen:
            stat, out = ctx.saved_tensors[4:6]
            check(_lib.lib().rfx_localstate_gen_bwd(_ptr(q), _ptr(k), _ptr(cont), _ptr(qd), _ptr(stat), _ptr(out), _ptr(g), B,
                                                    heads, ch, T, ndecay, _ptr(dq), _ptr(dk), _ptr(dc), _ptr(dqd), _stream()),
                  "rfx_localstate_gen_bwd")
        else:
            w = ctx.saved_tensors[4]
            check(_lib.lib().rfx_localstate_bwd(_ptr(q), _ptr(k), _ptr(cont), _ptr(qd), _ptr(w), _ptr(g), B, heads, ch, T,
                                                ndecay, _ptr(dq), _ptr(dk), _ptr(dc), _ptr(dqd), _stream()),
                  "rfx_localstate_bwd")
        return dq, dk, dc, dqd, None, None


def local_state_attention(q, k, content, query_decay, heads, ndecay):
    """q, k, content: (B, heads*ch, T); query_decay: (B, heads*ndecay, T) raw projections -> (B, heads*ch, T)."""
    return _LocalStateFn.apply(q, k, content, query_decay, heads, ndecay)


class _BlstmFrameFn(torch.autograd.Function):
    """(B, C, T) -> channel-major overlapping frames (1, C, width * B * nfr); rfx_blstm_frames modes 0 / 1."""

    @staticmethod
    def forward(ctx, x, nfr, width, stride):
        ops._req(x, "x")
        x = x.contiguous()
        B, Cc, T = x.shape
        h = torch.empty((1, Cc, width * B * nfr), device=x.device, dtype=torch.float32)
        check(_lib.lib().rfx_blstm_frames(_ptr(x), None, _ptr(h), B, Cc, T, nfr, width, stride, 0, _stream()), "rfx_blstm_frames")
        ctx.cfg = (B, Cc, T, nfr, width, stride)
        return h

    @staticmethod
    def backward(ctx, g):
        B, Cc, T, nfr, width, stride = ctx.cfg
        dx = torch.empty((B, Cc, T), device=g.device, dtype=torch.float32)
        check(_lib.lib().rfx_blstm_frames(_ptr(g.contiguous()), None, _ptr(dx), B, Cc, T, nfr, width, stride, 1, _stream()),
              "rfx_blstm_frames")
        return dx, None, None, None


class _BlstmUnframeFn(torch.autograd.Function):
    """frames (1, C, width * B * nfr) -> stitched (B, C, T) [+ skip]; rfx_blstm_frames modes 2 / 3."""

    @staticmethod
    def forward(ctx, h, skip, B, T, nfr, width, stride):
        ops._req(h, "h")
        h = h.contiguous()
        Cc = h.shape[1]
        out = torch.empty((B, Cc, T), device=h.device, dtype=torch.float32)
        sk = skip.contiguous() if skip is not None else None
        check(_lib.lib().rfx_blstm_frames(_ptr(h), _ptr(sk), _ptr(out), B, Cc, T, nfr, width, stride, 2, _stream()),
              "rfx_blstm_frames")
        ctx.cfg = (B, Cc, T, nfr, width, stride, skip is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        B, Cc, T, nfr, width, stride, has_skip = ctx.cfg
        g = g.contiguous()
        dh = torch.empty((1, Cc, width * B * nfr), device=g.device, dtype=torch.float32)
        check(_lib.lib().rfx_blstm_frames(_ptr(g), None, _ptr(dh), B, Cc, T, nfr, width, stride, 3, _stream()), "rfx_blstm_frames")
        return dh, (g if has_skip else None), None, None, None, None, None


def blstm_frame(x, nfr, width, stride):
    return _BlstmFrameFn.apply(x, nfr, width, stride)


def blstm_unframe(h, skip, B, T, nfr, width, stride):
    return _BlstmUnframeFn.apply(h, skip, B, T, nfr, width, stride)


class _DropoutFn(torch.autograd.Function):
    """F.dropout(x, p, training=True) with a counter-based mask (rfx_dropout): no mask tensor, backward re-draws it."""

    @staticmethod
    def forward(ctx, x, p, seed):
        ops._req(x, "x")
        xc = x.contiguous()
        out = torch.empty_like(xc)
        check(_lib.lib().rfx_dropout(_ptr(xc), _ptr(out), xc.numel(), float(p), int(seed), _stream()), "rfx_dropout")
        ctx.cfg = (float(p), int(seed))
        return out

    @staticmethod
    def backward(ctx, g):
        p, seed = ctx.cfg
        gc = g.contiguous()
        dx = torch.empty_like(gc)
        check(_lib.lib().rfx_dropout(_ptr(gc), _ptr(dx), gc.numel(), p, seed, _stream()), "rfx_dropout")
        return dx, None, None


def dropout(x, p, training=True):
    """The seed is drawn from torch's CPU generator (torch.manual_seed makes runs repeatable; no device sync)."""
    if not training or p <= 0.0:
        return x
    seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
    return _DropoutFn.apply(x, float(p), seed)


def linear(x, weight, bias):
    """x: (..., Cin) -> (..., Cout) as a 1x1 gather-GEMM over the flattened rows."""
    shp = x.shape
    x2 = x.reshape(1, -1, shp[-1]).transpose(1, 2)              # (1, Cin, rows) strided view
    y = ops.conv1d(x2, weight.unsqueeze(-1), bias)              # (1, Cout, rows)
    return y.transpose(1, 2).reshape(*shp[:-1], weight.shape[0])
