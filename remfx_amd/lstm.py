"""Bidirectional multi-layer LSTM on the HIP kernels (csrc/lstm.hip + the gather-GEMM).

Replaces ``nn.LSTM`` where the reference reaches it: torchaudio HDemucs' ``_BLSTM`` inside the DConv
branches (remfx/models.py:319) and Open-Unmix's 3-layer BiLSTM (remfx/models.py:297-298,
``cfg/model/umx.yaml``).  The ``nn.LSTM`` object stays the PARAMETER CONTAINER (same state_dict keys
``weight_ih_l0``, ``weight_hh_l0_reverse`` ...).

Layout: sequences are channel-major ``(1, C, T*Bn)`` with position ``p = t*Bn + b`` so that
  * the input projection of both directions is ONE 1x1 gather-GEMM  (8H x Cin),
  * the recurrence is one persistent launch per layer (rfx_lstm_fwd / rfx_lstm_bwd),
  * dX, dW_ih (+ bias row) are the conv dgrad / wgrad plans, and dW_hh is a wgrad whose input operand is
    the layer's own output shifted by one time step (convplan.shift_plan, +-Bn positions).
"""
import ctypes as C

import torch
import torch.nn.functional as F

from . import _lib, convplan, ops
from ._lib import check
from .ops import _ptr, _stream


def _pack_whh(w_hh, w_hh_r, Bn, train):
    H = w_hh.shape[1]
    L = _lib.lib()
    nb = L.rfx_lstm_pack_bytes(H)
    if nb <= 0:
        raise ValueError(f"LSTM hidden size {H} unsupported (multiple of 32, <= 512)")
    pack = torch.empty(2 * nb, device=w_hh.device, dtype=torch.uint8)
    # the single-workgroup form (csrc/lstm.hip) has its own fragment order: packed only for the shapes that take it
    fwd_l, bwd_l = L.rfx_lstm_local(H, Bn, ops.GEMM_PREC, 0), (train and L.rfx_lstm_local(H, Bn, ops.GEMM_PREC, 1))
    for d, w in enumerate((w_hh, w_hh_r)):
        wc = w.contiguous()
        dst = C.c_void_p(pack.data_ptr() + d * nb)
        if not (fwd_l and (bwd_l or not train)):           # some sweep still runs the cluster form
            check(L.rfx_lstm_pack(_ptr(wc), H, dst, _stream()), "rfx_lstm_pack")
        if fwd_l or bwd_l:
            check(L.rfx_lstm_pack_local(_ptr(wc), H, dst, _stream()), "rfx_lstm_pack_local")
    return pack


_WS = {}


def _workspace(device, H):
    """Per (device, H, stream) scratch for the cluster exchange; byte 0..3 = sticky error flag (see error_flag).  Keyed by the
    stream too: two recurrences of one H on two streams (a model whose time branch carries a BLSTM beside the frequency branch's)
    must not share exchange buffers and arrival counters."""
    key = (device.index, H, ops.raw_stream())
    ws = _WS.get(key)
    if ws is None:
        nb = _lib.lib().rfx_lstm_ws_bytes(H)
        if nb <= 0:
            raise ValueError(f"LSTM hidden size {H} unsupported (multiple of 32, <= 512)")
        ws = torch.zeros(nb, device=device, dtype=torch.uint8)
        _WS[key] = ws
    return ws


def error_flag():
    """True if any recurrence launch on this process timed out in a bounded spin (results invalid)."""
    return any(bool(ws[:4].view(torch.int32).item()) for ws in _WS.values())


def _whh_plan(out4, g4, H, d, Bn):
    """wgrad plan for dW_hh of direction d: in = out[d*H:(d+1)*H] at time t-1 (d=0) / t+1 (d=1)."""
    xs = out4[:, d * H:(d + 1) * H]
    gs = g4[:, d * 4 * H:(d + 1) * 4 * H]
    shift = -Bn if d == 0 else Bn
    key = ops._key("lstm_whh", xs.shape, xs.stride(), gs.stride(), shift)
    dp = ops._plans(key, out4.device, lambda: convplan.shift_plan(
        tuple(xs.shape), xs.stride(), 4 * H, shift, tuple(gs.shape), gs.stride()))
    return dp, xs, gs


class _LSTMLayerFn(torch.autograd.Function):
    """One bidirectional layer.  x: (1, Cin, P) -> (1, 2H, P)."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r, T, Bn):
        ops._req(x, "x")
        H, Cin = w_hh.shape[1], w_ih.shape[1]
        P = T * Bn
        if x.shape != (1, Cin, P):
            raise ValueError(f"lstm: expected (1, {Cin}, {P}) channel-major input, got {tuple(x.shape)}")
        x = x.contiguous()
        x4 = x.unsqueeze(2)
        if x.is_cuda and all(t.is_contiguous() and t.dtype == torch.float32 for t in (w_ih, w_ih_r, b_ih, b_hh, b_ih_r, b_hh_r)):
            wcat = torch.empty((8 * H, Cin, 1, 1), device=x.device, dtype=torch.float32)     # one launch instead of 2 cat + 2 add
            bcat = torch.empty((8 * H,), device=x.device, dtype=torch.float32)
            check(_lib.lib().rfx_lstm_cat_params(_ptr(w_ih), _ptr(w_ih_r), _ptr(b_ih), _ptr(b_hh), _ptr(b_ih_r), _ptr(b_hh_r), H, Cin,
                                                 _ptr(wcat), _ptr(bcat), _stream()), "rfx_lstm_cat_params")
        else:
            wcat = torch.cat([w_ih, w_ih_r]).view(8 * H, Cin, 1, 1)
            bcat = torch.cat([b_ih + b_hh, b_ih_r + b_hh_r])
        xp = ops.conv2d_forward(x4, wcat, bcat, (1, 1), (0, 0), (1, 1))           # (1, 8H, 1, P) == [2][4H][P]
        need = any(ctx.needs_input_grad)
        pack = _pack_whh(w_hh, w_hh_r, Bn, need)
        out = torch.empty((1, 2 * H, P), device=x.device, dtype=torch.float32)
        gates = torch.empty((2, 4 * H, P), device=x.device, dtype=torch.float32) if need else None
        cst = torch.empty((2, H, P), device=x.device, dtype=torch.float32) if need else None
        check(_lib.lib().rfx_lstm_fwd(_ptr(xp), _ptr(pack), T, Bn, H, _ptr(out), _ptr(gates), _ptr(cst),
                                      _ptr(_workspace(x.device, H)), ops.GEMM_PREC, _stream()),
              "rfx_lstm_fwd")
        ctx.prec = ops.GEMM_PREC
        if need:
            ctx.save_for_backward(x, wcat, pack, gates, cst, out)
            ctx.dims = (T, Bn, H, Cin)
            ctx.params = (w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r)    # identities for the GradSink lookup
        return out

    @staticmethod
    def backward(ctx, g):
        x, wcat, pack, gates, cst, out = ctx.saved_tensors
        T, Bn, H, Cin = ctx.dims
        P = T * Bn
        g = g.contiguous()
        dG = torch.empty((1, 8 * H, 1, P), device=g.device, dtype=torch.float32)
        check(_lib.lib().rfx_lstm_bwd(_ptr(g), _ptr(pack), _ptr(gates), _ptr(cst), T, Bn, H, _ptr(dG),
                                      _ptr(_workspace(g.device, H)), ctx.prec, _stream()),
              "rfx_lstm_bwd")
        x4 = x.unsqueeze(2)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv2d_dgrad(dG, wcat, tuple(x4.shape), x4.stride(), (1, 1), (0, 0), (1, 1)).squeeze(2)
        # the eight parameter gradients: with a GradSink armed (all eight recognised) they are computed on its side stream and
        # accumulated in place; nothing downstream in the backward chain needs them
        sink = ops.SINK
        tg = [sink.lookup(p) for p in ctx.params] if sink is not None else [None]
        sunk = all(t is not None for t in tg)
        stream = sink.stream_for_wgrad(x, dG, out) if sunk else torch.cuda.current_stream()
        with torch.cuda.stream(stream):
            dwcat, dbcat = ops.conv2d_wgrad(x4, dG, tuple(wcat.shape), (1, 1), (0, 0), (1, 1), True)
            dwcat = dwcat.view(8 * H, Cin)
            out4 = out.unsqueeze(2)
            dwhh = []
            for d in range(2):
                dp, xs, gs = _whh_plan(out4, dG, H, d, Bn)
                dap = ops.gemm_wgrad(dp, xs, gs)
                if sunk:                     # straight into the parameter's slice of the flat gradient buffer (no temporary, no add)
                    ops.unpack_add(dp, dap, tg[1 + 4 * d][1])
                    dwhh.append(None)
                    continue
                dw = ops.zeros((4 * H, H), g.device)
                ops.unpack_add(dp, dap, dw)
                dwhh.append(dw)
            db0, db1 = dbcat[:4 * H], dbcat[4 * H:]
            grads = (dwcat[:4 * H], dwhh[0], db0, db0, dwcat[4 * H:], dwhh[1], db1, db1)
            if sunk:                         # one launch instead of six add_ into the flat gradient buffer's views
                v = [t[1] for t in tg]
                check(_lib.lib().rfx_lstm_grad_scatter(_ptr(dwcat), _ptr(dbcat), H, Cin, _ptr(v[0]), _ptr(v[4]), _ptr(v[2]), _ptr(v[3]),
                                                       _ptr(v[6]), _ptr(v[7]), _stream()), "rfx_lstm_grad_scatter")
        if sunk:
            for i, _ in tg:
                sink.wrote(i)
            return (dx,) + (None,) * 10
        return (dx,) + grads + (None, None)


def blstm(module, x, T, Bn):
    """module: bidirectional ``nn.LSTM`` parameter container (batch_first irrelevant: the layout is explicit).
    x: (1, C, T*Bn) channel-major, position = t*Bn + b.  Returns (1, 2H, T*Bn)."""
    if not module.bidirectional:
        raise NotImplementedError("only bidirectional LSTMs are on the hot path (cfg/model/umx.yaml, HDemucs BLSTM)")
    if not module.bias or getattr(module, "proj_size", 0):
        raise NotImplementedError("lstm: bias=True, proj_size=0 only")
    h = x
    for layer in range(module.num_layers):
        p = [getattr(module, f"{n}_l{layer}{sfx}") for sfx in ("", "_reverse")
             for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        h = _LSTMLayerFn.apply(h, *p, T, Bn)
        if module.dropout > 0 and module.training and layer < module.num_layers - 1:
            from . import nnops
            h = nnops.dropout(h, module.dropout, True)
    return h
