"""Channels-last bf16 trunk of Hybrid Demucs' frequency branch (bf16 arithmetic mode = BASELINE config 3): the autograd nodes that
chain the kernels of csrc/cl_conv.hip / cl_wgrad.hip / cl_elem.hip over the norm-free layers of torchaudio HDemucs'
`freq_encoder` / `freq_decoder` (reference call site remfx/models.py:308,317).

Between those layers the activations stay (N, Fr, T, C) bf16.  What an elementwise pass used to do happens in the store of the GEMM
that produces its operand: GLU, GELU (+ the next skip add), and in the backward pass the GELU / GLU derivatives and the skip-gradient
add.  Three node types cover the branch:

  EncMidFn    DConv_i output -> rewrite 1x1 + GLU (+ frequency embedding) = skip e_i -> conv (8, 1) / 4 + GELU -> DConv_{i+1} input
  EncTailFn   DConv_J output -> rewrite 1x1 + GLU = skip e_J (channels-last) and the channel-major copy the normalised layers read
  FreqDecoderFn  deep output + skips e_J .. e_0 -> [rewrite 3x3 + GLU -> conv_tr (8, 1) / 4 + GELU + next skip] x J -> rewrite 3x3 + GLU

The DConv residual branches between them keep their own kernels (csrc/dconv.hip, norm.hip) on (N * Fr, C, T) fp32 samples; the
layout conversion at their ends is fused with the neighbouring elementwise step (rfx_cl_from_cm modes).  Weight gradients are
deterministic (fixed-order split reduction) and go straight into the GradSink slice when one is armed.
"""
import ctypes as C

import torch

from . import _lib, clast, ops
from ._lib import check

_FORMS = {}
_BUILD = {
    "glu": clast.form_conv_glu, "dgrad": clast.form_conv_dgrad, "s4": clast.form_conv_s4, "s4d": clast.form_conv_s4_dgrad,
    "tr": clast.form_convtr_s4, "trd": clast.form_convtr_s4_dgrad, "w": clast.wform_conv, "ws4": clast.wform_conv_s4,
    "wtr": clast.wform_convtr_s4,
    # the time branch: stride 4 along the position axis through folded views
    "s4f": clast.form_conv_s4_fold, "s4fd": clast.form_conv_s4_fold_dgrad, "trf": clast.form_convtr_fold,
    "trfd": clast.form_convtr_fold_dgrad, "ws4f": clast.wform_conv_s4_fold, "wtrf": clast.wform_convtr_fold,
    # the network's ends (1 - 2 channel tensors): 16-channel GEMMs over im2col operands, channel-major store
    "head": clast.form_head, "whead": clast.wform_head, "tail": clast.form_tail_tr, "taild": clast.form_tail_dgrad, "wtail": clast.wform_tail,
}


def _fold(t, k=4):
    """(N, 1, L, C) -> its folded view (N, 1, L / k, k C): k consecutive positions as channels."""
    N, A, L, Cc = t.shape
    return t.view(N, A, L // k, k * Cc)


def form(kind, *a):
    key = (kind,) + a
    f = _FORMS.get(key)
    if f is None:
        f = _BUILD[kind](*a)
        _FORMS[key] = f
    return f


_PACKS = {}


def packed(f, w):
    """MFMA fragments of weight w for GEMM form f, re-packed when the weight changed (ops.weights_changed / torch version).  An
    entry holds a reference to the tensor it was packed from: its storage cannot be freed and handed to another weight under the
    same address while the entry lives (the keys are addresses)."""
    key = (id(f), w.data_ptr(), ops.raw_stream())
    try:
        ver = w._version
    except RuntimeError:
        ver = None
    e = _PACKS.get(key)
    if e is not None and ver is not None and e[0] == ver and e[1] == ops._WEIGHT_EPOCH[0] and e[3] is f:
        return e[2]
    ap = clast.pack(f, w.detach())
    if ver is not None:
        if len(_PACKS) > 2048:
            _PACKS.clear()
        _PACKS[key] = (ver, ops._WEIGHT_EPOCH[0], ap, f, w.detach())
    return ap


def clear_packs():
    _PACKS.clear()


def rowsum(x, A, Cc, out, scale=1.0, accumulate=False):
    """out[a][c] (+)= scale * sum_{n, b} x[n][a][b][c] for a contiguous channels-last x; A == 1 folds the rows into n."""
    N, XA, B, XC = x.shape
    ct = clast.cl_tensor(x)
    if A == 1 and XA != 1:
        if not x.is_contiguous():
            raise ValueError("rowsum: folding rows needs a contiguous tensor")
        N, ct.ns = N * XA, ct.as_
    G = max(1, min(N, -(-2048 // A)))
    partial = torch.empty((G, A, Cc), device=x.device, dtype=torch.float32)
    check(_lib.lib().rfx_cl_rowsum(C.byref(ct), N, A, B, Cc, G, float(scale), C.c_void_p(partial.data_ptr()), C.c_void_p(out.data_ptr()),
                                   int(accumulate), C.c_void_p(ops.raw_stream())), "rfx_cl_rowsum")
    return out


def _wgrad(f, p, q, N, OA, IA, B, w, b, bias_src=None):
    """Weight (and bias) gradient of one layer: into the armed GradSink's slices on its side stream (returns None, None), else fresh
    tensors for autograd.  bias_src: the channels-last tensor whose position sum is the bias gradient when the GEMM form does not
    carry it (transposed convolutions: the bias gradient sums Q, not P)."""
    sink = ops.SINK
    tw = sink.lookup(w) if sink is not None else None
    tb = sink.lookup(b) if (tw is not None and b is not None) else None
    if tw is not None and (b is None or tb is not None):
        with torch.cuda.stream(sink.stream_for_wgrad(p, q)):
            clast.wgrad(f, p, q, N, OA, IA, B, tw[1], tb[1] if (f.bias and b is not None) else _scratch_bias(f, p), accumulate=True)
            if b is not None and not f.bias:
                rowsum(bias_src, 1, b.numel(), tb[1], accumulate=True)
        sink.wrote(tw[0])
        if b is not None:
            sink.wrote(tb[0])
        return None, None
    dw = torch.empty(w.shape, device=w.device, dtype=torch.float32)
    db = torch.empty(b.shape, device=w.device, dtype=torch.float32) if b is not None else None
    clast.wgrad(f, p, q, N, OA, IA, B, dw, db if (f.bias and b is not None) else _scratch_bias(f, p))
    if b is not None and not f.bias:
        rowsum(bias_src, 1, b.numel(), db)
    return dw, db


def _scratch_bias(f, p):
    return torch.empty(f.M, device=p.device, dtype=torch.float32) if f.bias else None


def _as_ncab(x3, Bn, A):
    """(Bn * A, C, T) sample-major tensor -> (Bn, C, A, T) strided view (no copy)."""
    NA, Cc, T = x3.shape
    return x3.view(Bn, A, Cc, T).permute(0, 2, 1, 3)


class EncMidFn(torch.autograd.Function):
    """d_i (Bn * A_i, C_i, T) fp32 -> e_i channels-last (Bn, A_i, T, C_i), y_{i+1} (Bn * A_i / 4, 2 C_i, T) fp32."""

    @staticmethod
    def forward(ctx, d, rw_w, rw_b, cv_w, cv_b, emb_rows, Bn, y_cl, fold):
        """fold: the time branch -- A = 1, the stride-4 convolution runs along the position axis (folded views)."""
        d_is_cl = d.dim() == 4                      # the DConv branch in front ran on channels-last samples (cldconv)
        if d_is_cl:
            _, A, T, Cc = d.shape
            d_cl = d if d.is_contiguous() else d.contiguous()
        else:
            NA, Cc, T = d.shape
            A = NA // Bn
            d_cl = clast.from_cm(_as_ncab(d if d.is_contiguous() else d.contiguous(), Bn, A))
        C1 = cv_w.shape[0]
        A1, T1 = (A, T // 4) if fold else (A // 4, T)
        dev = d.device
        train = any(ctx.needs_input_grad)
        zab = clast.empty(Bn, A, T, 2 * Cc, dev) if train else None
        e = clast.empty(Bn, A, T, Cc, dev)
        fg = form("glu", 2 * Cc, Cc, 1, 1)
        clast.conv(fg, packed(fg, rw_w), d_cl, Bn, A, T, A, "glu", bias=rw_b, out0=zab, out1=e,
                   rowadd=emb_rows.contiguous() if emb_rows is not None else None)
        z1 = clast.empty(Bn, A1, T1, C1, dev) if train else None
        y1 = clast.empty(Bn, A1, T1, C1, dev)
        if fold:
            fs = form("s4f", C1, Cc)
            clast.conv(fs, packed(fs, cv_w), _fold(e), Bn, 1, T1, 1, "gelu", bias=cv_b, out0=z1, out1=y1)
        else:
            fs = form("s4", C1, Cc)
            clast.conv(fs, packed(fs, cv_w), e, Bn, A, T, A1, "gelu", bias=cv_b, out0=z1, out1=y1)
        if y_cl:                                    # the next DConv branch takes channels-last samples
            y_out = y1
        else:
            y_out = torch.empty((Bn * A1, C1, T1), device=dev, dtype=torch.float32)
            clast.to_cm(y1, out=_as_ncab(y_out, Bn, A1))
        if train:
            ctx.save_for_backward(d_cl, zab, e, z1, rw_w, cv_w)
            ctx.refs = (rw_b, cv_b)
            ctx.geom = (Bn, A, T, Cc, C1, A1, T1, emb_rows is not None, d_is_cl, y_cl, fold)
        return e, y_out

    @staticmethod
    def backward(ctx, g_e, g_y):
        d_cl, zab, e, z1, rw_w, cv_w = ctx.saved_tensors
        rw_b, cv_b = ctx.refs
        Bn, A, T, Cc, C1, A1, T1, has_emb, d_is_cl, y_cl, fold = ctx.geom
        dev = d_cl.device
        # gradient of the next DConv's input (-> channels-last) times gelu'(z1)
        if y_cl:
            dz1 = clast.dgelu(g_y if g_y.is_contiguous() else g_y.contiguous(), z1)
        else:
            dz1 = clast.empty(Bn, A1, T1, C1, dev)
            clast.from_cm(_as_ncab(g_y if g_y.is_contiguous() else g_y.contiguous(), Bn, A1), out=dz1, aux=z1, mode="dgelu")
        g_e = (g_e if g_e.is_contiguous() else g_e.contiguous()) if g_e is not None else None
        if fold:
            dcw, dcb = _wgrad(form("ws4f", C1, Cc), dz1, _fold(e), Bn, 1, 1, T1, cv_w, cv_b)
            # conv input gradient (written through the folded view) + skip gradient; the GLU backward is a pass of its own here:
            # folded rows are (sub-position, channel), the stored [a | b] pairs are not where the fused store looks for them
            fd = form("s4fd", C1, Cc)
            v = clast.empty(Bn, A, T, Cc, dev)
            clast.conv(fd, packed(fd, cv_w), dz1, Bn, 1, T1, 1, "store", out0=_fold(v), res=_fold(g_e) if g_e is not None else None)
            dzab = clast.dglu(v, zab)
        else:
            dcw, dcb = _wgrad(form("ws4", C1, Cc), dz1, e, Bn, A1, A, T, cv_w, cv_b)
            # conv input gradient + skip gradient, GLU backward against the stored [a | b]
            fd = form("s4d", C1, Cc)
            dzab = clast.empty(Bn, A, T, 2 * Cc, dev)
            v = clast.empty(Bn, A, T, Cc, dev) if has_emb else None
            clast.conv(fd, packed(fd, cv_w), dz1, Bn, A1, T, A1 + 1, "dglu", out0=dzab, out1=v, aux0=zab, res=g_e, OAo=A)
        demb = None
        if has_emb:
            demb = torch.empty((A, Cc), device=dev, dtype=torch.float32)
            rowsum(v, A, Cc, demb)
        drw, drb = _wgrad(form("w", 2 * Cc, Cc, 1, 1), dzab, d_cl, Bn, A, A, T, rw_w, rw_b)
        fr = form("dgrad", 2 * Cc, Cc, 1, 1)
        dd_cl = clast.empty(Bn, A, T, Cc, dev)
        clast.conv(fr, packed(fr, rw_w), dzab, Bn, A, T, A, "store", out0=dd_cl)
        if d_is_cl:
            dd = dd_cl
        else:
            dd = torch.empty((Bn * A, Cc, T), device=dev, dtype=torch.float32)
            clast.to_cm(dd_cl, out=_as_ncab(dd, Bn, A))
        return dd, drw, drb, dcw, dcb, demb, None, None, None


class EncTailFn(torch.autograd.Function):
    """d_J (Bn * A, C, T) fp32 -> e_J channels-last (the skip) and e_J (Bn, C, A, T) fp32 (what the normalised layer above reads)."""

    @staticmethod
    def forward(ctx, d, rw_w, rw_b, Bn):
        d_is_cl = d.dim() == 4
        if d_is_cl:
            _, A, T, Cc = d.shape
            d_cl = d if d.is_contiguous() else d.contiguous()
        else:
            NA, Cc, T = d.shape
            A = NA // Bn
            d_cl = clast.from_cm(_as_ncab(d if d.is_contiguous() else d.contiguous(), Bn, A))
        dev = d.device
        train = any(ctx.needs_input_grad)
        zab = clast.empty(Bn, A, T, 2 * Cc, dev) if train else None
        e = clast.empty(Bn, A, T, Cc, dev)
        fg = form("glu", 2 * Cc, Cc, 1, 1)
        clast.conv(fg, packed(fg, rw_w), d_cl, Bn, A, T, A, "glu", bias=rw_b, out0=zab, out1=e)
        e_cm = clast.to_cm(e)
        if train:
            ctx.save_for_backward(d_cl, zab, rw_w)
            ctx.refs = (rw_b,)
            ctx.geom = (Bn, A, T, Cc, d_is_cl)
        return e, e_cm

    @staticmethod
    def backward(ctx, g_e, g_cm):
        d_cl, zab, rw_w = ctx.saved_tensors
        (rw_b,) = ctx.refs
        Bn, A, T, Cc, d_is_cl = ctx.geom
        dev = d_cl.device
        if g_cm is None:
            raise RuntimeError("EncTailFn: the channel-major output carries no gradient")
        if g_cm.stride(3) != 1:
            g_cm = g_cm.contiguous()
        dzab = clast.empty(Bn, A, T, 2 * Cc, dev)
        clast.from_cm(g_cm, out=dzab, res=g_e, aux=zab, mode="dglu")
        drw, drb = _wgrad(form("w", 2 * Cc, Cc, 1, 1), dzab, d_cl, Bn, A, A, T, rw_w, rw_b)
        fr = form("dgrad", 2 * Cc, Cc, 1, 1)
        dd_cl = clast.empty(Bn, A, T, Cc, dev)
        clast.conv(fr, packed(fr, rw_w), dzab, Bn, A, T, A, "store", out0=dd_cl)
        if d_is_cl:
            dd = dd_cl
        else:
            dd = torch.empty((Bn * A, Cc, T), device=dev, dtype=torch.float32)
            clast.to_cm(dd_cl, out=_as_ncab(dd, Bn, A))
        return dd, drw, drb, None


class FreqDecoderFn(torch.autograd.Function):
    """x (Bn, C_J, A_J, T) fp32 from the normalised layers, skips e_J .. e_0 channels-last, then per layer j = J .. 1 the weights
    (rewrite w, b, conv_tr w, b) and the last layer's rewrite (w, b).  Returns y_0 = GLU(rewrite_0(.)) as (Bn, C_0, A_0, T) fp32."""

    @staticmethod
    def forward(ctx, x, nsk, fold, tail, *rest):
        """fold: the time branch (A = 1, 1 x 3 rewrites, the transposed convolutions through folded views).  tail: the parameters end
        with the LAST layer's transposed convolution (w, b; C -> 1 | 2 channels): the node returns its channel-major fp32 output."""
        skips, params = rest[:nsk], rest[nsk:]
        J = nsk - 1
        Bn, CJ, AJ, T = x.shape
        KA = 1 if fold else 3
        dev = x.device
        train = any(ctx.needs_input_grad)
        if x.stride(3) != 1:
            x = x.contiguous()
        xin = clast.from_cm(x, res=skips[0])
        saved = []
        A, Cc = AJ, CJ
        for k in range(J):                                      # layer j = J - k
            rw_w, rw_b, ct_w, ct_b = params[4 * k:4 * k + 4]
            Cn = ct_w.shape[1]
            fg = form("glu", 2 * Cc, Cc, KA, 3)
            zab = clast.empty(Bn, A, T, 2 * Cc, dev) if train else None
            y = clast.empty(Bn, A, T, Cc, dev)
            clast.conv(fg, packed(fg, rw_w), xin, Bn, A, T, A, "glu", bias=rw_b, out0=zab, out1=y)
            if fold:
                ft = form("trf", Cc, Cn)
                zt = clast.empty(Bn, 1, 4 * T, Cn, dev) if train else None
                nxt = clast.empty(Bn, 1, 4 * T, Cn, dev)
                clast.conv(ft, packed(ft, ct_w), y, Bn, 1, T, 1, "gelu", bias=ct_b, out0=_fold(zt) if train else None, out1=_fold(nxt),
                           aux0=_fold(skips[k + 1]))
                saved += [xin, zab, y, zt]
                xin, T, Cc = nxt, 4 * T, Cn
                continue
            ft = form("tr", Cc, Cn)
            zt = clast.empty(Bn, 4 * A, T, Cn, dev) if train else None
            nxt = clast.empty(Bn, 4 * A, T, Cn, dev)
            clast.conv(ft, packed(ft, ct_w), y, Bn, A, T, A + 1, "gelu", bias=ct_b, out0=zt, out1=nxt, aux0=skips[k + 1], OAo=4 * A)
            saved += [xin, zab, y, zt]
            xin, A, Cc = nxt, 4 * A, Cn
        rw_w, rw_b = params[4 * J:4 * J + 2]
        fg = form("glu", 2 * Cc, Cc, KA, 3)
        zab = clast.empty(Bn, A, T, 2 * Cc, dev) if train else None
        y0 = clast.empty(Bn, A, T, Cc, dev)
        clast.conv(fg, packed(fg, rw_w), xin, Bn, A, T, A, "glu", bias=rw_b, out0=zab, out1=y0)
        saved += [xin, zab]
        if tail:
            tw, tb = params[4 * J + 2], params[4 * J + 3]
            Cs = tw.shape[1]
            if fold:
                out = torch.empty((Bn, Cs, 1, 4 * T), device=dev, dtype=torch.float32)
                ftl = form("trf", Cc, Cs)
                clast.conv(ftl, packed(ftl, tw), y0, Bn, 1, T, 1, "store_cm", bias=tb, cm_out=out, cm_fold=True)
            else:
                out = torch.empty((Bn, Cs, 4 * A, T), device=dev, dtype=torch.float32)
                ftl = form("tail", Cc, Cs)
                clast.conv(ftl, packed(ftl, tw), y0, Bn, A, T, A + 1, "store_cm", bias=tb, cm_out=out, OAo=4 * A)
            saved += [y0]
        else:
            out = clast.to_cm(y0)
        if train:
            ctx.save_for_backward(*saved, *[p for p in params])
            ctx.nsaved = len(saved)
            ctx.geom = (Bn, AJ, CJ, x.shape[3], J, fold, tail)
        return out

    @staticmethod
    def backward(ctx, g):
        Bn, AJ, CJ, TJ, J, fold, tail = ctx.geom
        KA = 1 if fold else 3
        saved, params = ctx.saved_tensors[:ctx.nsaved], ctx.saved_tensors[ctx.nsaved:]
        dev = g.device
        if g.stride(3) != 1:
            g = g.contiguous()
        grads_p = [None] * len(params)
        grads_sk = [None] * (J + 1)
        # layer 0: GLU backward fused with the layout conversion of the incoming gradient
        A, T, Cc = (AJ, TJ * 4 ** J, CJ // 2 ** J) if fold else (AJ * 4 ** J, TJ, CJ // 2 ** J)
        xin, zab = saved[4 * J], saved[4 * J + 1]
        dzab = clast.empty(Bn, A, T, 2 * Cc, dev)
        if tail:
            # the last transposed convolution: its output gradient gathered per input position (16 channels = 8 taps x Cs), the input
            # gradient as a 16-channel GEMM with the rewrite's GLU backward in its store
            y0 = saved[4 * J + 2]
            tw, tb = params[4 * J + 2], params[4 * J + 3]
            Cs = tw.shape[1]
            g16 = clast.im2col_s4(g, A, T, fold)
            grads_p[4 * J + 2], _ = _wgrad(form("wtail", Cc, Cs), y0, g16, Bn, A, A, T, tw, None)
            sink = ops.SINK
            tbs = sink.lookup(tb) if sink is not None else None
            dtb = ops.channel_sum(g)
            if tbs is not None:
                tbs[1].add_(dtb)
                sink.wrote(tbs[0])
            else:
                grads_p[4 * J + 3] = dtb
            ftd = form("taild", Cc, Cs)
            clast.conv(ftd, packed(ftd, tw), g16, Bn, A, T, A, "dglu", out0=dzab, aux0=zab)
        else:
            clast.from_cm(g, out=dzab, aux=zab, mode="dglu")
        for k in range(J, -1, -1):                              # rewrite of layer j = J - k, walking up from layer 0 (k = J)
            rw_w, rw_b = params[4 * k], params[4 * k + 1]
            grads_p[4 * k], grads_p[4 * k + 1] = _wgrad(form("w", 2 * Cc, Cc, KA, 3), dzab, xin, Bn, A, A, T, rw_w, rw_b)
            fd = form("dgrad", 2 * Cc, Cc, KA, 3)
            gx = clast.empty(Bn, A, T, Cc, dev)
            if k == 0:
                clast.conv(fd, packed(fd, rw_w), dzab, Bn, A, T, A, "store", out0=gx)
                grads_sk[0] = gx
                break
            # this layer's input = gelu(z') + skip of the layer above: skip gradient = gx, pre-activation gradient = gx * gelu'(z')
            xin_u, zab_u, y_u, zt_u = saved[4 * (k - 1):4 * (k - 1) + 4]
            dzt = clast.empty(Bn, A, T, Cc, dev)
            clast.conv(fd, packed(fd, rw_w), dzab, Bn, A, T, A, "dgelu", out0=gx, out1=dzt, aux0=zt_u)
            grads_sk[k] = gx
            ct_w, ct_b = params[4 * (k - 1) + 2], params[4 * (k - 1) + 3]
            if fold:
                Cu, Tu = 2 * Cc, T // 4
                grads_p[4 * (k - 1) + 2], grads_p[4 * (k - 1) + 3] = _wgrad(form("wtrf", Cu, Cc), y_u, _fold(dzt), Bn, 1, 1, Tu, ct_w, ct_b,
                                                                            bias_src=dzt.view(Bn * 64, 1, T // 64, Cc))
                ftd = form("trfd", Cu, Cc)
                dzab = clast.empty(Bn, 1, Tu, 2 * Cu, dev)
                clast.conv(ftd, packed(ftd, ct_w), _fold(dzt), Bn, 1, Tu, 1, "dglu", out0=dzab, aux0=zab_u)
                xin, T, Cc = xin_u, Tu, Cu
                continue
            Cu, Au = 2 * Cc, A // 4
            grads_p[4 * (k - 1) + 2], grads_p[4 * (k - 1) + 3] = _wgrad(form("wtr", Cu, Cc), y_u, dzt, Bn, Au, A, T, ct_w, ct_b, bias_src=dzt)
            ftd = form("trd", Cu, Cc)
            dzab = clast.empty(Bn, Au, T, 2 * Cu, dev)
            clast.conv(ftd, packed(ftd, ct_w), dzt, Bn, A, T, Au, "dglu", out0=dzab, aux0=zab_u)
            xin, A, Cc = xin_u, Au, Cu
        gx_cm = clast.to_cm(grads_sk[0])
        return (gx_cm, None, None, None, *grads_sk, *grads_p)


class HeadGeluFn(torch.autograd.Function):
    """z (Bn, C, A, T) bf16 channel-major -- the first encoder layer's convolution output, whose two input channels keep it off the
    channels-last kernels -- -> gelu(z) as channels-last samples (Bn, A, T, C); backward: g * gelu'(z) back in z's layout."""

    @staticmethod
    def forward(ctx, z):
        ctx.save_for_backward(z)
        return clast.from_cm(z, mode="gelu")

    @staticmethod
    def backward(ctx, g):
        (z,) = ctx.saved_tensors
        gz = torch.empty_like(z)
        clast.to_cm(g if g.is_contiguous() else g.contiguous(), out=gz, aux16=z)
        return gz


def head_gelu(z):
    if z.dtype != torch.bfloat16 or not z.is_contiguous():
        raise ValueError("head_gelu: contiguous bf16 (N, C, A, T)")
    return HeadGeluFn.apply(z)


def _w4(w):
    """Conv1d / ConvTranspose1d weights as the 4-D tensors the 2-D forms index (a view: the GradSink recognises the storage)."""
    return w if w.dim() == 4 else w.unsqueeze(2)


def enc_mid(d, rewrite, conv_next, emb_rows, Bn, y_cl=False, fold=False):
    return EncMidFn.apply(d, _w4(rewrite.weight), rewrite.bias, _w4(conv_next.weight), conv_next.bias, emb_rows, Bn, y_cl, fold)


def enc_tail(d, rewrite, Bn):
    return EncTailFn.apply(d, _w4(rewrite.weight), rewrite.bias, Bn)


def freq_decoder(x, skips, layers, fold=False, tail=False):
    """skips: [e_J, ..., e_0]; layers: the _HDecLayer modules of layers J .. 0; fold: the time branch's decoder (x: (B, C, 1, L));
    tail: the last layer's transposed convolution too (returns its channel-major output instead of its input)."""
    params = []
    for m in layers[:-1]:
        params += [_w4(m.rewrite.weight), m.rewrite.bias, _w4(m.conv_tr.weight), m.conv_tr.bias]
    params += [_w4(layers[-1].rewrite.weight), layers[-1].rewrite.bias]
    if tail:
        params += [_w4(layers[-1].conv_tr.weight), layers[-1].conv_tr.bias]
    return FreqDecoderFn.apply(x, len(skips), fold, tail, *skips, *params)


class HeadConvFn(torch.autograd.Function):
    """The frequency branch's first convolution (2 spectrogram channels -> C, (8, 1) / 4, padding 2) + GELU as a 16-channel GEMM over the
    im2col of its channel-major fp32 input: x (Bn, 2, 4 A, T) -> gelu(z) channels-last (Bn, A, T, C).  The input carries no gradient."""

    @staticmethod
    def forward(ctx, x, w, b, along_b, coef=None):
        """along_b: the time branch's first convolution (1 waveform channel, stride along the samples): x (Bn, 1, 1, L) -> (Bn, 1, L / 4, C).
        coef = (a, b): x is the UN-standardised frame-major spectrum (Bn, T, 4 A, 2) and the operand is a x + b (round 6: clast.im2col_fm)."""
        if coef is not None:
            Bn, T, IA, Cs = x.shape
            A = IA // 4
            x16 = clast.im2col_fm(x, coef[0], coef[1])
        else:
            Bn, Cs, IA, T = x.shape
            A, T = (IA, T // 4) if along_b else (IA // 4, T)
            x16 = clast.im2col_s4(x, A, T, along_b)
        Cc = w.shape[0]
        dev = x.device
        train = any(ctx.needs_input_grad[1:3])
        f = form("head", Cc, Cs)
        z = clast.empty(Bn, A, T, Cc, dev) if train else None
        y = clast.empty(Bn, A, T, Cc, dev)
        clast.conv(f, packed(f, w), x16, Bn, A, T, A, "gelu", bias=b, out0=z, out1=y)
        if train:
            ctx.save_for_backward(x16, z, w)
            ctx.refs = (b,)
        return y

    @staticmethod
    def backward(ctx, g):
        x16, z, w = ctx.saved_tensors
        (b,) = ctx.refs
        Bn, A, T, Cc = z.shape
        dz = clast.dgelu(g if g.is_contiguous() else g.contiguous(), z)
        dw, db = _wgrad(form("whead", Cc, w.shape[1]), dz, x16, Bn, A, A, T, w, b)
        return None, dw, db, None, None


def head_conv(x, conv, along_b=False):
    return HeadConvFn.apply(x, _w4(conv.weight), conv.bias, along_b)


def head_conv_fm(spec_fm, a, b, conv):
    """The frequency branch's first convolution + GELU on the frame-major spectrum (Bn, frames, bins, 2), standardisation a x + b folded
    into the operand: -> (Bn, bins / 4, frames, C) channels-last."""
    return HeadConvFn.apply(spec_fm, _w4(conv.weight), conv.bias, False, (a, b))
