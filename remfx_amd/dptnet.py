"""asteroid.models.DPTNet as RemFX configures it (reference remfx/models.py:327-344 `DPTNetModel`,
cfg/model/dptnet.yaml) on the HIP kernels.

Same constructor keywords and state_dict keys as asteroid's class (``encoder.filterbank._filters``,
``masker.layers.{r}.{0,1}.{mha.in_proj_weight, recurrent.weight_ih_l0, norm_mha.gamma, ...}``, ``masker.first_out.1.weight`` ...),
so RemFX checkpoints load strictly; the torch modules inside are PARAMETER CONTAINERS.  Arithmetic:
  * free filterbank encoder / decoder = strided conv1d / conv_transpose1d on the gather-GEMM kernels,
  * global layer norm (gLN) = the GroupNorm(1, C) kernels with eps 1e-8,
  * multi-head attention = three 1x1 GEMMs + the streaming attention kernels (rfx_mha_fwd / bwd, csrc/attention.hip) + one 1x1 GEMM,
  * the BiLSTM feed-forward = the wave-cluster recurrence kernels (remfx_amd/lstm.py) between two 1x1 GEMMs,
  * PReLU / tanh / sigmoid / relu / adds = elementwise kernels;
chunking (unfold / 50 % overlap-add) and the (batch, channel, time) <-> sequence re-layouts are tensor views and copies.
asteroid is absent from the reference tree and this image: parity unpinned, oracle/ref_dptnet.py is the fp32 restatement.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, lstm, nnops, ops
from ._lib import check
from .ops import _ptr, _stream

EPS = 1e-8


class GlobLN(nn.Module):
    def __init__(self, channel_size):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(channel_size))
        self.beta = nn.Parameter(torch.zeros(channel_size))

    def forward(self, x):                                     # (B, C, S): statistics over (C, S) per item
        return nnops.group_norm(x.contiguous(), 1, self.gamma, self.beta, EPS)


class FreeFB(nn.Module):
    def __init__(self, n_filters, kernel_size, stride):
        super().__init__()
        self.n_filters, self.kernel_size, self.stride = n_filters, kernel_size, stride
        self._filters = nn.Parameter(torch.ones(n_filters, 1, kernel_size))
        for p in self.parameters():
            nn.init.xavier_normal_(p)


class _Coder(nn.Module):
    def __init__(self, fb):
        super().__init__()
        self.filterbank = fb


class _MhaFn(torch.autograd.Function):
    """softmax(k^T q / sqrt(ch)) v per head on the streaming attention kernels; q, k, v: (B, heads * ch, T)."""

    @staticmethod
    def forward(ctx, q, k, v, heads):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        B, E, T = q.shape
        ch = E // heads
        out = torch.empty_like(q)
        stat = torch.empty((B * heads * T, 4), device=q.device, dtype=torch.float32)
        check(_lib.lib().rfx_mha_fwd(_ptr(q), _ptr(k), _ptr(v), B, heads, ch, T, _ptr(stat), _ptr(out), _stream()), "rfx_mha_fwd")
        ctx.save_for_backward(q, k, v, stat, out)
        ctx.cfg = (B, heads, ch, T)
        return out

    @staticmethod
    def backward(ctx, g):
        q, k, v, stat, out = ctx.saved_tensors
        B, heads, ch, T = ctx.cfg
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        check(_lib.lib().rfx_mha_bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(stat), _ptr(out), _ptr(g.contiguous()), B, heads, ch, T,
                                     _ptr(dq), _ptr(dk), _ptr(dv), _stream()), "rfx_mha_bwd")
        return dq, dk, dv, None


class _PReLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, slope):
        x = x.contiguous()
        C = slope.numel()
        N, L = (x.shape[0], x[0].numel()) if C == 1 else (x.shape[0], x[0, 0].numel())
        y = torch.empty_like(x)
        check(_lib.lib().rfx_prelu_fwd(_ptr(x), _ptr(slope), _ptr(y), N, C, L, _stream()), "rfx_prelu_fwd")
        ctx.save_for_backward(x, slope)
        ctx.cfg = (N, C, L)
        return y

    @staticmethod
    def backward(ctx, g):
        x, slope = ctx.saved_tensors
        N, C, L = ctx.cfg
        gx, gs = torch.empty_like(x), torch.empty_like(slope)
        ws = torch.empty(N * C, device=x.device, dtype=torch.float64)
        check(_lib.lib().rfx_prelu_bwd(_ptr(x), _ptr(g.contiguous()), _ptr(slope), _ptr(gx), _ptr(ws), _ptr(gs), N, C, L, _stream()),
              "rfx_prelu_bwd")
        return gx, gs


class ImprovedTransformedLayer(nn.Module):
    """x + MHA(x) -> gLN -> x + Linear(relu(BiLSTM(x))) -> gLN on (batch, channels, seq)."""

    def __init__(self, embed_dim, n_heads, dim_ff, dropout=0.0, bidirectional=True):
        super().__init__()
        self.n_heads = n_heads
        self.mha = nn.MultiheadAttention(embed_dim, n_heads, dropout=dropout)
        self.recurrent = nn.LSTM(embed_dim, dim_ff, bidirectional=bidirectional, batch_first=True)
        self.linear = nn.Linear(2 * dim_ff if bidirectional else dim_ff, embed_dim)
        self.norm_mha = GlobLN(embed_dim)
        self.norm_ff = GlobLN(embed_dim)

    def forward(self, x):
        Bp, E, S = x.shape
        w, b = self.mha.in_proj_weight, self.mha.in_proj_bias
        q, k, v = (ops.conv1d(x, w[i * E:(i + 1) * E].unsqueeze(-1), b[i * E:(i + 1) * E]) for i in range(3))
        att = _MhaFn.apply(q, k, v, self.n_heads)
        out = ops.conv1d(att, self.mha.out_proj.weight.unsqueeze(-1), self.mha.out_proj.bias)
        x = self.norm_mha(nnops.add(out, x))
        # BiLSTM feed-forward in the recurrence kernels' channel-major layout: position = t * Bp + b
        seq = x.permute(1, 2, 0).reshape(1, E, S * Bp)
        h = lstm.blstm(self.recurrent, seq, S, Bp)
        h = ops.activation(h, "relu")
        h = ops.conv1d(h, self.linear.weight.unsqueeze(-1), self.linear.bias)
        out = h.view(E, S, Bp).permute(2, 0, 1)
        return self.norm_ff(nnops.add(out.contiguous(), x))


class DPTransformer(nn.Module):
    def __init__(self, in_chan, n_src, n_heads=4, ff_hid=256, chunk_size=100, hop_size=None, n_repeats=6):
        super().__init__()
        if in_chan % n_heads:
            raise NotImplementedError("DPTransformer: in_chan must be a multiple of n_heads (asteroid adds an input Linear otherwise)")
        self.in_chan, self.n_src, self.chunk_size = in_chan, n_src, chunk_size
        self.hop_size = hop_size if hop_size is not None else chunk_size // 2
        self.in_norm = GlobLN(in_chan)
        self.layers = nn.ModuleList([nn.ModuleList([ImprovedTransformedLayer(in_chan, n_heads, ff_hid),
                                                    ImprovedTransformedLayer(in_chan, n_heads, ff_hid)])
                                     for _ in range(n_repeats)])
        self.first_out = nn.Sequential(nn.PReLU(), nn.Conv2d(in_chan, n_src * in_chan, 1))
        self.net_out = nn.Sequential(nn.Conv1d(in_chan, in_chan, 1), nn.Tanh())
        self.net_gate = nn.Sequential(nn.Conv1d(in_chan, in_chan, 1), nn.Sigmoid())

    def _fold(self, u, frames):
        """overlap-add of (B, C, chunk, n_chunks) back to (B, C, frames), divided by chunk / hop (asteroid DualPathProcessing.fold)."""
        B, C, K, n = u.shape
        if 2 * self.hop_size != K:                              # general hop: torch's col2im (layout + adds)
            out = F.fold(u.reshape(B, C * K, n), (frames, 1), kernel_size=(K, 1), padding=(K, 0), stride=(self.hop_size, 1))
            return out.reshape(B, C, frames) / (float(K) / self.hop_size)
        # hop = chunk / 2: the even chunks tile the padded axis, the odd chunks tile it shifted by one hop -> two views + one add
        Lp = frames + 2 * K
        ev, od = u[..., 0::2], u[..., 1::2]
        e = F.pad(ev.permute(0, 1, 3, 2).reshape(B, C, -1), (0, Lp - ev.shape[-1] * K))
        o = F.pad(od.permute(0, 1, 3, 2).reshape(B, C, -1), (self.hop_size, Lp - self.hop_size - od.shape[-1] * K))
        tot = nnops.add(e, o)[..., K:K + frames].contiguous()
        half = torch.full((B * C,), 0.5, device=u.device)
        return nnops.row_affine(tot.reshape(B * C, frames), half, torch.zeros_like(half)).view(B, C, frames)

    def forward(self, w):
        w = self.in_norm(w)
        B, C, frames = w.shape
        K = self.chunk_size
        u = F.unfold(w.unsqueeze(-1), kernel_size=(K, 1), padding=(K, 0), stride=(self.hop_size, 1)).reshape(B, C, K, -1)
        n = u.shape[-1]
        for intra, inter in self.layers:
            v = u.transpose(1, -1).reshape(B * n, K, C).transpose(1, -1).contiguous()
            v = intra(v)
            u = v.reshape(B, n, C, K).transpose(1, -1).transpose(1, 2)
            v = u.transpose(1, 2).reshape(B * K, C, n)
            v = inter(v)
            u = v.reshape(B, K, C, n).transpose(1, 2)
        u = _PReLUFn.apply(u.contiguous(), self.first_out[0].weight)
        out = ops.conv2d(u, self.first_out[1].weight, self.first_out[1].bias)
        out = out.reshape(B * self.n_src, self.in_chan, K, n)
        out = self._fold(out, frames)
        a = ops.activation(ops.conv1d(out, self.net_out[0].weight, self.net_out[0].bias), "tanh")
        g = ops.activation(ops.conv1d(out, self.net_gate[0].weight, self.net_gate[0].bias), "sigmoid")
        return ops.activation(nnops.mul(a, g).reshape(B, self.n_src, self.in_chan, frames), "relu")


class DPTNet(nn.Module):
    def __init__(self, n_src, n_heads=4, ff_hid=256, chunk_size=100, hop_size=None, n_repeats=6, norm_type="gLN",
                 ff_activation="relu", encoder_activation="relu", mask_act="relu", bidirectional=True, dropout=0, in_chan=None,
                 fb_name="free", kernel_size=16, n_filters=64, stride=8, sample_rate=8000, **fb_kwargs):
        super().__init__()
        if ((fb_name, norm_type, ff_activation, encoder_activation, mask_act) != ("free", "gLN", "relu", "relu", "relu")
                or not bidirectional or dropout):
            raise NotImplementedError("DPTNet: the cfg/model/dptnet.yaml form (free filterbank, gLN, relu, bidirectional, dropout 0)")
        if in_chan is not None and in_chan != n_filters:
            raise ValueError("in_chan must equal the filterbank's n_filters")
        self.n_src, self.stride = n_src, stride
        self.encoder = _Coder(FreeFB(n_filters, kernel_size, stride))
        self.decoder = _Coder(FreeFB(n_filters, kernel_size, stride))
        self.masker = DPTransformer(n_filters, n_src, n_heads, ff_hid, chunk_size, hop_size, n_repeats)

    def forward(self, wav):                                    # (B, T) -> (B, n_src, T)
        ops._req(wav, "wav")
        x = wav.unsqueeze(1)
        tf = ops.activation(ops.conv1d(x, self.encoder.filterbank._filters, None, self.stride), "relu")
        masks = self.masker(tf)
        B, S, N, Fr = masks.shape
        masked = nnops.mul(masks, tf.unsqueeze(1).expand(B, S, N, Fr).contiguous())
        dec = ops.conv_transpose1d(masked.reshape(B * S, N, Fr), self.decoder.filterbank._filters, None, self.stride)
        dec = dec.reshape(B, S, -1)
        T = wav.shape[-1]
        return F.pad(dec, (0, T - dec.shape[-1])) if dec.shape[-1] < T else dec[..., :T]
