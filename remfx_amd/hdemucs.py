"""Hybrid Demucs (torchaudio.models.HDemucs as RemFX configures it: reference
remfx/models.py:307-324, cfg/model/demucs.yaml:11-16) on the HIP kernels.

Same constructor signature, sub-module names and state_dict keys as the upstream
class (freq_encoder.{i}.conv / norm1 / rewrite / norm2 / dconv.layers.{d}.{idx},
time_encoder, freq_decoder.{j}.conv_tr / norm2 / rewrite / norm1, time_decoder,
freq_emb.embedding.weight), so RemFX checkpoints load strictly.
nn.Conv* / nn.GroupNorm / nn.LSTM objects are PARAMETER CONTAINERS only; the
arithmetic goes through remfx_amd.ops / stft / nnops (HIP kernels behind the C ABI).

Data layout in HBM: spectrogram branch (B, C, Fr, T) with T (256 frames) contiguous,
time branch (B, C, L) with L contiguous -- the gather-GEMM kernels take the position
axis from the contiguous dimension, so both branches are read coalesced.
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import clchain, cldconv, lstm, nnops, ops, stft

# bf16 mode: the norm-free frequency layers run on the channels-last bf16 trunk (remfx_amd/clchain.py); RFX_CL_TRUNK=0 keeps the
# channel-major kernels of rounds 1-4 for same-box A/B runs
CL_TRUNK = os.environ.get("RFX_CL_TRUNK", "1") != "0"
CL_TIME = os.environ.get("RFX_CL_TIME", "1") != "0"         # ... and the time branch's norm-free layers (folded-view forms, clast.py)
CL_ENDS = os.environ.get("RFX_CL_ENDS", "1") != "0"         # ... and the 1 - 2 channel convolutions at the network's ends as im2col GEMMs
FM_ENDS = os.environ.get("RFX_FM_ENDS", "1") != "0"           # round 6: frame-major spectrum on both sides of the U-Net (A/B: 0)
TWO_STREAMS = os.environ.get("RFX_TWO_STREAMS", "1") != "0"   # the time branch on a second high-priority stream (its layers 0-3 and the
                                                               # frequency layers 0-3 are independent between the input and layer 4)
_TIME_STREAMS = {}


def _data_parallel():
    """More than one rank AND no gradient sink to order the exchange: ddp.GradSync orders a bucket's all-reduce behind every stream
    the sink has seen gradient writes on (ops.GradSink.write_streams: compute, weight-gradient, time-branch), so the multi-rank step
    runs the same stream configuration as the single-rank one (round 6).  Without an armed sink (GradSink.MODE = "off") the collective
    only follows the stream it is issued from, and the time branch stays on the compute stream."""
    if not (torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1):
        return False
    return torch.is_grad_enabled() and ops.SINK is None


def _dev_env(name, default):
    """Hazard-hunt / measurement switches (they serialise streams, skip whole DConv branches or re-enable a known-bad statistics
    path): honoured only under RFX_DEV=1, and then announced -- a leaked variable must not silently change results."""
    v = os.environ.get(name)
    if v is None or v == default:
        return default
    if os.environ.get("RFX_DEV") != "1":
        import warnings
        warnings.warn(f"{name}={v} ignored: development switch, set RFX_DEV=1 to honour it")
        return default
    import warnings
    warnings.warn(f"RFX_DEV=1: development switch {name}={v} is ACTIVE and changes what the network computes or how it is ordered")
    return v


_XSUB = int(_dev_env("RFX_XSYNC_SUB", "0"))
_CUR = [None, None]
_XIDX = int(_dev_env("RFX_XSYNC_IDX", "-1"))
_XSYNC = int(_dev_env("RFX_XSYNC", "0"))      # dev: serialisation points of the two-stream hazard hunt


def _time_stream(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _TIME_STREAMS.get(idx)
    if st is None:
        st = torch.cuda.Stream(device=idx, priority=-1)
        _TIME_STREAMS[idx] = st
    return st


CL_DCONV = os.environ.get("RFX_CL_DCONV", "1") != "0"
CL_TIME_DCONV = os.environ.get("RFX_CL_TIME_DCONV", "1") != "0"    # ... the time branch's too (whole-clip GroupNorm: the kernels run in passes)       # ... and their DConv branches on the fused channels-last kernels (cldconv.py)


class _ScaledEmbedding(nn.Module):
    def __init__(self, num_embeddings, embedding_dim, scale=10.0, smooth=False):
        super().__init__()
        self.embedding = nn.Embedding(num_embeddings, embedding_dim)
        if smooth:
            w = torch.cumsum(self.embedding.weight.data, dim=0)
            w = w / torch.arange(1, num_embeddings + 1).sqrt()[:, None]
            self.embedding.weight.data[:] = w
        self.embedding.weight.data /= scale
        self.scale = scale

    def table(self):
        """(num_embeddings, dim) scaled table == forward(arange(n))."""
        return self.embedding.weight * self.scale


# partial-sum slots per sample for the GEMM-epilogue GroupNorm statistics (spreads same-address fp64 atomics)
_STAT_SLOTS = 16


class _LayerScale(nn.Module):
    def __init__(self, channels, init=0.0):
        super().__init__()
        self.scale = nn.Parameter(torch.full((channels,), float(init)))


class _BLSTM(nn.Module):
    def __init__(self, dim, layers=2, skip=True):
        super().__init__()
        self.max_steps = 200
        self.lstm = nn.LSTM(bidirectional=True, num_layers=layers, hidden_size=dim, input_size=dim)
        self.linear = nn.Linear(2 * dim, dim)
        self.skip = skip

    def forward(self, x):
        B, C, T = x.shape
        if T > self.max_steps:                       # overlapping frames of 200 steps at stride 100, central parts stitched
            width, stride = self.max_steps, self.max_steps // 2
            nfr = math.ceil(T / stride)
        else:
            width, stride, nfr = T, T, 1
        Bn = B * nfr
        h = nnops.blstm_frame(x, nfr, width, stride)                 # (1, C, width * Bn): position = t * Bn + b * nfr + k
        h = lstm.blstm(self.lstm, h, width, Bn)
        h = ops.conv1d(h, self.linear.weight.unsqueeze(-1), self.linear.bias)
        return nnops.blstm_unframe(h, x if self.skip else None, B, T, nfr, width, stride)


class _LocalState(nn.Module):
    def __init__(self, channels, heads=4, ndecay=4):
        super().__init__()
        self.heads, self.ndecay = heads, ndecay
        self.content = nn.Conv1d(channels, channels, 1)
        self.query = nn.Conv1d(channels, channels, 1)
        self.key = nn.Conv1d(channels, channels, 1)
        self.query_decay = nn.Conv1d(channels, heads * ndecay, 1)
        self.query_decay.weight.data *= 0.01
        self.query_decay.bias.data[:] = -2
        self.proj = nn.Conv1d(channels, channels, 1)

    def forward(self, x):
        q = ops.conv1d(x, self.query.weight, self.query.bias)
        k = ops.conv1d(x, self.key.weight, self.key.bias)
        c = ops.conv1d(x, self.content.weight, self.content.bias)
        qd = ops.conv1d(x, self.query_decay.weight, self.query_decay.bias)
        res = nnops.local_state_attention(q, k, c, qd, self.heads, self.ndecay)
        return nnops.add(x, ops.conv1d(res, self.proj.weight, self.proj.bias))


# The C >= 192 DConv branches took their GroupNorm statistics from the producing GEMM's epilogue (fp64 atomics into a zero-filled slot
# buffer): those lose contributions while a second stream keeps the machine busy.  Default now: the GroupNorm kernel
# computes them from the stored 16-bit tensor with per-chunk stores (what autocast's GroupNorm sees); 1 = the epilogue form (A/B).
DCONV_EPI_STATS = _dev_env("RFX_DCONV_EPI_STATS", "0") != "0"
_ST_MODE = int(_dev_env("RFX_ST_MODE", "0"))     # dev: how the statistics buffer of the channel-major DConv is zeroed (hazard hunt)


def _stat_buf(x):
    if _ST_MODE == 1:
        return torch.zeros((x.shape[0], _STAT_SLOTS, 2), device=x.device, dtype=torch.float64)
    st = ops.zeros((x.shape[0], _STAT_SLOTS, 2), x.device, torch.float64)
    if _ST_MODE == 2:                                   # a second fill: does a late write-back of the first one matter?
        ops._lib.check(ops._lib.lib().rfx_zero(ops._ptr(st), st.numel() * 8, ops._stream()), "rfx_zero")
    return st


_DCONV_DBG = None       # dev (scripts/probes/dconv_steps.py): a list collects per-sample checksums after every step of the channel-major path
_DBG_SKIP = {int(v) for v in _dev_env('RFX_DBG_SKIP_DCONV_C', '').split(',') if v}   # measurement only


class _DConv(nn.Module):
    def __init__(self, channels, compress=4, depth=2, init=1e-4, norm_type="group_norm", attn=False,
                 heads=4, ndecay=4, lstm=False, kernel_size=3):
        super().__init__()
        hidden = int(channels / compress)
        self.layers = nn.ModuleList()
        self.spec = []
        for d in range(depth):
            dil = 2 ** d
            mods = [nn.Conv1d(channels, hidden, kernel_size, dilation=dil, padding=dil * (kernel_size // 2)),
                    nn.GroupNorm(1, hidden), nn.GELU(), nn.Conv1d(hidden, 2 * channels, 1),
                    nn.GroupNorm(1, 2 * channels), nn.GLU(1), _LayerScale(channels, init)]
            if attn:
                mods.insert(3, _LocalState(hidden, heads=heads, ndecay=ndecay))
            if lstm:
                mods.insert(3, _BLSTM(hidden, layers=2, skip=True))
            self.layers.append(nn.Sequential(*mods))
            self.spec.append((dil, dil * (kernel_size // 2), lstm, attn))

    def cl_ok(self, positions=256):
        """The fused channels-last kernels (csrc/cl_dconv.hip) take every depth-layer of this branch, forward and backward, for samples
        of `positions` positions (256: a frequency row; the time branch: a whole clip, in 256-position tiles)."""
        for seq, (dil, pad, lstm, attn) in zip(self.layers, self.spec):
            m = list(seq)
            if lstm or attn or m[0].kernel_size[0] != 3 or dil not in (1, 2) or m[1].eps != m[4].eps or m[1].num_groups != 1:
                return False
            if not cldconv.supported(m[0].in_channels, m[0].out_channels, True, positions):
                return False
        return True

    def forward_cl(self, x):
        """x: (B, Fr, T, C) channels-last bf16 samples."""
        for seq, (dil, pad, lstm, attn) in zip(self.layers, self.spec):
            m = list(seq)
            x = cldconv.dconv_layer(x, m[0], m[1], m[3], m[4], m[6].scale, dil)
        return x

    def forward(self, x):
        if _DBG_SKIP and self.layers[0][0].in_channels in _DBG_SKIP:
            return x
        for seq, (dil, pad, lstm, attn) in zip(self.layers, self.spec):
            mods = list(seq)
            if (not lstm and not attn and mods[1].eps == mods[4].eps and mods[1].num_groups == 1 and
                    nnops.dconv_layer_fused_ok(x, mods[0].out_channels, mods[0].kernel_size[0], dil,
                                               torch.is_grad_enabled() and (x.requires_grad or mods[0].weight.requires_grad))):
                # frequency-branch samples of (C, 256): the whole depth-layer in one launch per direction (csrc/dconv.hip)
                x = nnops.dconv_layer(x, mods[0], mods[1], mods[3], mods[4], mods[6].scale, dil)
                continue
            st = _stat_buf(x) if DCONV_EPI_STATS else None   # GN(1, C) statistics out of the GEMM epilogue (A/B), else from the stored tensor
            y, x = ops.conv1d_fork(x, mods[0].weight, mods[0].bias, 1, pad, dil, stat_sums=st, out_bf16=True)   # statistics come out of the GEMM epilogue
            if _DCONV_DBG is not None:
                _DCONV_DBG.append(("conv1", y.float().abs().sum(dim=(1, 2)), st.sum(dim=1).clone() if st is not None else None))
            y = nnops.group_norm(y, 1, mods[1].weight, mods[1].bias, mods[1].eps, mode="gelu", sums=st)
            if _DCONV_DBG is not None:
                _DCONV_DBG.append(("gn1", y.float().abs().sum(dim=(1, 2)), None))
            i = 3
            if lstm:
                y = mods[i](y); i += 1
            if attn:
                y = mods[i](y); i += 1
            st = _stat_buf(x) if DCONV_EPI_STATS else None
            # the 2C-channel tensor is read only by the GroupNorm + GLU kernel (and, in backward, its gradient only by GEMMs):
            # 16-bit storage in the bf16 mode
            y = ops.conv1d(y, mods[i].weight, mods[i].bias, stat_sums=st, out_bf16=True)
            if _DCONV_DBG is not None:
                _DCONV_DBG.append(("conv2", y.float().abs().sum(dim=(1, 2)), st.sum(dim=1).clone() if st is not None else None))
            x = nnops.group_norm(y, 1, mods[i + 1].weight, mods[i + 1].bias, mods[i + 1].eps,
                                 mode="glu_scale_res", res=x, scale=mods[i + 3].scale, sums=st)
            if _DCONV_DBG is not None:
                _DCONV_DBG.append(("gn2", x.float().abs().sum(dim=(1, 2)), None))
        return x


def _norm(groups, ch, on):
    return nn.GroupNorm(groups, ch) if on else nn.Identity()


def _norm_act(m, x, act):
    """act(norm(x)) with act in {none, gelu, glu}; norm may be nn.Identity."""
    if isinstance(m, nn.GroupNorm):
        return nnops.group_norm(x, m.num_groups, m.weight, m.bias, m.eps, mode=act)
    if act == "gelu":
        return nnops.gelu(x)
    if act == "glu":
        return nnops.glu(x, 1)
    return x


ENC_Z16 = True      # bf16 mode: encoder conv outputs of the norm-free layers stored in 16 bits (bench.py --no-enc-z16 is the A/B)
ENC_Z16_TIME = True # ... of the time branch too (bench.py --no-enc-z16-time)


class _HEncLayer(nn.Module):
    def __init__(self, chin, chout, kernel_size=8, stride=4, norm_groups=4, empty=False, freq=True,
                 norm_type="group_norm", context=0, dconv_kw=None, pad=True):
        super().__init__()
        padv = kernel_size // 4 if pad else 0
        self.freq, self.kernel_size, self.stride, self.empty, self.pad = freq, kernel_size, stride, empty, padv
        self.context = context
        norm = norm_type == "group_norm"
        if freq:
            self.conv = nn.Conv2d(chin, chout, (kernel_size, 1), (stride, 1), (padv, 0))
        else:
            self.conv = nn.Conv1d(chin, chout, kernel_size, stride, padv)
        self.norm1 = _norm(norm_groups, chout, norm)
        if empty:
            self.rewrite, self.norm2, self.dconv = nn.Identity(), nn.Identity(), nn.Identity()
        else:
            klass = nn.Conv2d if freq else nn.Conv1d
            self.rewrite = klass(chout, 2 * chout, 1 + 2 * context, 1, context)
            self.norm2 = _norm(norm_groups, 2 * chout, norm)
            self.dconv = _DConv(chout, **(dconv_kw or {}))

    def forward(self, x, inject=None, fork=False):
        """fork=True (x is the previous layer's output, which is also a skip connection): returns (out, alias of x); the caller
        keeps the ALIAS as the skip, so both gradients of x reach this layer's conv backward together and the skip gradient is
        added in the input-gradient GEMM's store (ops.ConvFork2dFn) instead of a separate accumulation pass over the tensor."""
        alias, want_pair = None, fork
        if not self.freq and x.dim() == 4:
            x = x.reshape(x.shape[0], -1, x.shape[-1])
            fork = False
        if self.freq:
            # bf16 mode, layers without a norm: the conv output is read only by the GELU pass (and its backward), its gradient only
            # by the input- / weight-gradient GEMMs, which round to bf16 anyway -- both are STORED in 16 bits (what torch autocast
            # stores for a conv output).
            z16 = ENC_Z16 and (not self.empty) and inject is None and not isinstance(self.norm1, nn.GroupNorm)
            if fork:
                y, alias = ops.conv2d_fork(x, self.conv.weight, self.conv.bias, (self.stride, 1), (self.pad, 0), out_bf16=z16)
            else:
                y = ops.conv2d(x, self.conv.weight, self.conv.bias, (self.stride, 1), (self.pad, 0), out_bf16=z16)
        else:
            if x.shape[-1] % self.stride:
                x = F.pad(x, (0, self.stride - x.shape[-1] % self.stride))
                fork = False
            # time branch, same rule: the strided plans' weight-gradient kernel (gemm_wgrad_bf_kernel<.., G16>) reads the 16-bit gradient
            z16 = ENC_Z16 and ENC_Z16_TIME and (not self.empty) and not isinstance(self.norm1, nn.GroupNorm)
            if fork:
                y, alias = ops.conv1d_fork(x, self.conv.weight, self.conv.bias, self.stride, self.pad, out_bf16=z16)
            else:
                y = ops.conv1d(x, self.conv.weight, self.conv.bias, self.stride, self.pad, out_bf16=z16)
        out = self._rest(y, inject)
        return (out, alias) if want_pair else out

    def head(self, x, cl=False, ends=True):
        """conv + GELU of a norm-free frequency layer as (B * Fr, C, T) samples, the DConv branch's input (channels-last trunk);
        cl: as (B, Fr, T, C) channels-last bf16 samples (the fused channels-last DConv kernels)."""
        if cl and ends and x.shape[1] <= 2 and x.shape[2] % 4 == 0 and not x.requires_grad:
            return clchain.head_conv(x, self.conv)          # 16-channel GEMM over the im2col of the 2-channel spectrogram
        y = ops.conv2d(x, self.conv.weight, self.conv.bias, (self.stride, 1), (self.pad, 0), out_bf16=ENC_Z16 or cl)
        if cl:
            return clchain.head_gelu(y)
        y = ops.activation_to(y, "gelu", (0, 2, 1))
        B, C, Fr, T = y.shape
        return y.permute(0, 2, 1, 3).reshape(-1, C, T)

    def head_t(self, x):
        """conv + GELU of a norm-free TIME layer, (B, C, L / stride) fp32: the DConv branch's input (channels-last trunk)."""
        y = ops.conv1d(x, self.conv.weight, self.conv.bias, self.stride, self.pad, out_bf16=ENC_Z16 and ENC_Z16_TIME)
        return nnops.gelu(y)

    def _rest(self, y, inject):
        if self.empty:
            return y
        if inject is not None:
            if inject.dim() == 3 and y.dim() == 4:
                inject = inject[:, :, None]
            y = nnops.add(y, inject)
        if self.freq and not isinstance(self.norm1, nn.GroupNorm):
            # GELU written straight into the (B, Fr, C, T) order the DConv branch views as (B*Fr, C, T): no permuted copy,
            # and the backward reads the branch's gradient from that order (ops.ActToFn / rfx_act_rows)
            y = ops.activation_to(y, "gelu", (0, 2, 1))
        else:
            y = _norm_act(self.norm1, y, "gelu")
        if self.freq:
            B, C, Fr, T = y.shape
            y = self.dconv(y.permute(0, 2, 1, 3).reshape(-1, C, T))
            y = y.view(B, Fr, C, T).permute(0, 2, 1, 3)
            c = self.context
            if not isinstance(self.norm2, nn.GroupNorm):              # no norm between conv and GLU: GLU in the GEMM store
                return ops.conv2d_glu(y, self.rewrite.weight, self.rewrite.bias, (1, 1), (c, c))
            z = ops.conv2d(y, self.rewrite.weight, self.rewrite.bias, (1, 1), (c, c))
        else:
            y = self.dconv(y)
            if not isinstance(self.norm2, nn.GroupNorm):
                return ops.conv1d_glu(y, self.rewrite.weight, self.rewrite.bias, 1, self.context)
            z = ops.conv1d(y, self.rewrite.weight, self.rewrite.bias, 1, self.context)
        return _norm_act(self.norm2, z, "glu")


class _HDecLayer(nn.Module):
    def __init__(self, chin, chout, last=False, kernel_size=8, stride=4, norm_groups=1, empty=False,
                 freq=True, norm_type="group_norm", context=1, dconv_kw=None, pad=True):
        super().__init__()
        self.pad = (kernel_size - stride) // 2 if pad else 0
        self.last, self.freq, self.chin, self.empty = last, freq, chin, empty
        self.stride, self.kernel_size, self.context = stride, kernel_size, context
        norm = norm_type == "group_norm"
        if freq:
            self.conv_tr = nn.ConvTranspose2d(chin, chout, (kernel_size, 1), (stride, 1))
        else:
            self.conv_tr = nn.ConvTranspose1d(chin, chout, kernel_size, stride)
        self.norm2 = _norm(norm_groups, chout, norm)
        if empty:
            self.rewrite, self.norm1 = nn.Identity(), nn.Identity()
        else:
            klass = nn.Conv2d if freq else nn.Conv1d
            self.rewrite = klass(chin, 2 * chin, 1 + 2 * context, 1, context)
            self.norm1 = _norm(norm_groups, 2 * chin, norm)

    def forward(self, x, skip, length, next_skip=None, skip_added=False):
        """next_skip: the skip tensor the NEXT decoder layer will add to this layer's output; where this layer ends in a plain GELU
        it is added in the same pass (ops.activation_add) and `self.fused_next_add` tells the caller to pass skip_added=True on."""
        self.fused_next_add = False
        if self.freq and x.dim() == 3:
            x = x.view(x.shape[0], self.chin, -1, x.shape[-1])
        if not self.empty:
            if not skip_added:
                x = nnops.add(x, skip)
            c = self.context
            if not isinstance(self.norm1, nn.GroupNorm):              # GLU in the GEMM store
                y = (ops.conv2d_glu(x, self.rewrite.weight, self.rewrite.bias, (1, 1), (c, c)) if self.freq else
                     ops.conv1d_glu(x, self.rewrite.weight, self.rewrite.bias, 1, c))
            else:
                if self.freq:
                    r = ops.conv2d(x, self.rewrite.weight, self.rewrite.bias, (1, 1), (c, c))
                else:
                    r = ops.conv1d(x, self.rewrite.weight, self.rewrite.bias, 1, c)
                y = _norm_act(self.norm1, r, "glu")
        else:
            y = x
            if skip is not None:
                raise ValueError("skip must be None when empty is true.")
        act = "none" if self.last else "gelu"
        if isinstance(self.norm2, nn.GroupNorm):
            # upstream order: z = norm2(conv_tr(y)) over the FULL transposed-conv output, crop afterwards -- the
            # GroupNorm statistics include the border samples the crop drops (deep layers only: index >= norm_starts)
            if self.freq:
                zf = ops.conv_transpose2d(y, self.conv_tr.weight, self.conv_tr.bias, (self.stride, 1), (1, 1))
                zf = _norm_act(self.norm2, zf, act)
                z = zf[:, :, self.pad:zf.shape[2] - self.pad, :] if self.pad else zf
            else:
                zf = ops.conv_transpose1d(y, self.conv_tr.weight, self.conv_tr.bias, self.stride, 1)
                z = _norm_act(self.norm2, zf, act)[..., self.pad:self.pad + length]
            return z, y
        if self.freq:      # Identity norm: crop [pad : -pad] along Fr folded into the transposed-conv plan
            full = (y.shape[2] - 1) * self.stride + self.kernel_size
            z = ops.conv_transpose2d(y, self.conv_tr.weight, self.conv_tr.bias, (self.stride, 1), (1, 1),
                                     (self.pad, 0), (full - 2 * self.pad, y.shape[3]))
        else:              # crop [pad : pad + length]
            z = ops.conv_transpose1d(y, self.conv_tr.weight, self.conv_tr.bias, self.stride, 1, self.pad, length)
        if (next_skip is not None and act == "gelu" and not isinstance(self.norm2, nn.GroupNorm) and z.is_contiguous()
                and next_skip.shape == z.shape):
            self.fused_next_add = True
            return ops.activation_add(z, next_skip, "gelu"), y
        z = _norm_act(self.norm2, z, act)
        return z, y


class HDemucs(nn.Module):
    def __init__(self, sources, audio_channels=2, channels=48, growth=2, nfft=4096, depth=6, freq_emb=0.2,
                 emb_scale=10, emb_smooth=True, kernel_size=8, time_stride=2, stride=4, context=1,
                 context_enc=0, norm_starts=4, norm_groups=4, dconv_depth=2, dconv_comp=4, dconv_attn=4,
                 dconv_lstm=4, dconv_init=1e-4):
        super().__init__()
        self.depth, self.nfft, self.audio_channels, self.sources = depth, nfft, audio_channels, list(sources)
        self.kernel_size, self.context, self.stride, self.channels = kernel_size, context, stride, channels
        self.hop_length = nfft // 4
        self.freq_emb = None
        self.freq_encoder, self.freq_decoder = nn.ModuleList(), nn.ModuleList()
        self.time_encoder, self.time_decoder = nn.ModuleList(), nn.ModuleList()
        chin, chin_z = audio_channels, audio_channels * 2
        chout, chout_z = channels, channels
        freqs = nfft // 2
        for index in range(depth):
            lstm, attn = index >= dconv_lstm, index >= dconv_attn
            norm_type = "group_norm" if index >= norm_starts else "none"
            freq = freqs > 1
            stri, ker = stride, kernel_size
            if not freq:
                ker, stri = time_stride * 2, time_stride
            pad, last_freq = True, False
            if freq and freqs <= kernel_size:
                ker, pad, last_freq = freqs, False, True
            kw = {"kernel_size": ker, "stride": stri, "freq": freq, "pad": pad, "norm_type": norm_type,
                  "norm_groups": norm_groups,
                  "dconv_kw": {"lstm": lstm, "attn": attn, "depth": dconv_depth, "compress": dconv_comp,
                               "init": dconv_init}}
            kwt = dict(kw, freq=False, kernel_size=kernel_size, stride=stride, pad=True)
            kw_dec = dict(kw)
            if last_freq:
                chout_z = max(chout, chout_z)
                chout = chout_z
            self.freq_encoder.append(_HEncLayer(chin_z, chout_z, context=context_enc, **kw))
            if freq:
                if last_freq and nfft == 2048:
                    kwt["stride"], kwt["kernel_size"] = 2, 4
                self.time_encoder.append(_HEncLayer(chin, chout, context=context_enc, empty=last_freq, **kwt))
            if index == 0:
                chin = audio_channels * len(self.sources)
                chin_z = chin * 2
            self.freq_decoder.insert(0, _HDecLayer(chout_z, chin_z, last=index == 0, context=context, **kw_dec))
            if freq:
                self.time_decoder.insert(0, _HDecLayer(chout, chin, empty=last_freq, last=index == 0,
                                                       context=context, **kwt))
            chin, chin_z = chout, chout_z
            chout, chout_z = int(growth * chout), int(growth * chout_z)
            if freq:
                freqs = 1 if freqs <= kernel_size else freqs // stride
            if index == 0 and freq_emb:
                self.freq_emb = _ScaledEmbedding(freqs, chin_z, smooth=emb_smooth, scale=emb_scale)
                self.freq_emb_scale = freq_emb
        for m in self.modules():                         # init-time rescale, reference = 0.1
            if isinstance(m, (nn.Conv1d, nn.ConvTranspose1d, nn.Conv2d, nn.ConvTranspose2d)):
                s = (m.weight.std().detach() / 0.1) ** 0.5
                m.weight.data /= s
                if m.bias is not None:
                    m.bias.data /= s

    def forward_use_order(self):
        """The parameters in the order forward() first uses their layers: time / frequency encoders interleaved, the frequency embedding
        behind layer 0, then the decoders interleaved.  Backward completes their gradients in the reverse order; optim.FlatParams lays the
        flat buffers out this way so that ddp.GradSync's buckets finish one after the other during backward (registration order puts the
        frequency embedding and every time layer behind all frequency layers: the first bucket then waits for the end of backward and
        holds the in-order all-reduces of all the others back)."""
        out, offset = [], self.depth - len(self.time_decoder)
        for idx, enc in enumerate(self.freq_encoder):
            if idx < len(self.time_encoder):
                out += list(self.time_encoder[idx].parameters())
            out += list(enc.parameters())
            if idx == 0 and self.freq_emb is not None:
                out += list(self.freq_emb.parameters())
        for idx, dec in enumerate(self.freq_decoder):
            out += list(dec.parameters())
            if idx >= offset:
                out += list(self.time_decoder[idx - offset].parameters())
        seen = {id(p) for p in out}
        return out + [p for p in self.parameters() if id(p) not in seen]

    def _cl_layers(self, le, device):
        """How many leading frequency layers take the channels-last bf16 trunk (0: none).  Conditions: bf16 arithmetic mode, whole
        256-frame tiles, norm-free layers of the standard geometry (conv (8, 1) / 4 pad 2, 1x1 encoder rewrite, 3x3 decoder rewrite),
        channel counts in whole 16-channel K steps."""
        if not CL_TRUNK or ops.GEMM_PREC != 2 or device.type != "cuda" or le % 256 or le <= 0:
            return 0
        n, rows = 0, self.nfft // 2
        for i, enc in enumerate(self.freq_encoder):
            dec = self.freq_decoder[self.depth - 1 - i]
            ok = (enc.freq and not enc.empty and enc.context == 0 and enc.kernel_size == 8 and enc.stride == 4 and enc.pad == 2
                  and not isinstance(enc.norm1, nn.GroupNorm) and not isinstance(enc.norm2, nn.GroupNorm)
                  and enc.conv.out_channels % 16 == 0 and rows % 4 == 0
                  and dec.freq and not dec.empty and dec.context == 1 and dec.kernel_size == 8 and dec.stride == 4 and dec.pad == 2
                  and not isinstance(dec.norm1, nn.GroupNorm) and not isinstance(dec.norm2, nn.GroupNorm))
            if not ok:
                break
            n, rows = n + 1, rows // 4
        return n if 2 <= n < self.depth else 0

    def _cl_layers_time(self, length, device, Lc):
        """The time branch follows the frequency branch onto the channels-last trunk (same layer count) when its layers have the same
        norm-free geometry and every level is whole 256-position tiles."""
        if not CL_TIME or Lc == 0 or len(self.time_encoder) <= Lc or length % (4 ** Lc * 256):
            return 0
        for i in range(Lc):
            enc, dec = self.time_encoder[i], self.time_decoder[len(self.time_decoder) - 1 - i]
            ok = (not enc.freq and not enc.empty and enc.context == 0 and enc.kernel_size == 8 and enc.stride == 4 and enc.pad == 2
                  and not isinstance(enc.norm1, nn.GroupNorm) and not isinstance(enc.norm2, nn.GroupNorm)
                  and enc.conv.out_channels % 16 == 0
                  and not dec.freq and not dec.empty and dec.context == 1 and dec.kernel_size == 8 and dec.stride == 4 and dec.pad == 2
                  and not isinstance(dec.norm1, nn.GroupNorm) and not isinstance(dec.norm2, nn.GroupNorm))
            if not ok:
                return 0
        return Lc

    def _time_encoder_step(self, idx, Lt, B, saved_t, lengths_t, xt, samp_t, len_t):
        """Layer idx of the time encoder (on whatever stream is current).  Returns (xt, samp_t, len_t, inject): inject = the tensor the
        frequency branch's layer reads (the empty time layer's output), else None."""
        inject = None
        tenc = self.time_encoder[idx]
        if idx < Lt:
            # time branch on the channels-last trunk: as the frequency layers, with A = 1 and the stride along positions
            lengths_t.append(len_t)
            len_t = len_t // 4
            tdcl = CL_DCONV and CL_TIME_DCONV and tenc.dconv.cl_ok(len_t)      # len_t: this layer's clip length after its convolution
            if idx == 0:
                if tdcl and CL_ENDS and xt.shape[1] == 1 and not xt.requires_grad:
                    samp_t = clchain.head_conv(xt.unsqueeze(2), tenc.conv, along_b=True)     # (B, 1, L / 4, C) channels-last
                else:
                    samp_t, tdcl = tenc.head_t(xt), False
            dt_ = tenc.dconv.forward_cl(samp_t) if tdcl else tenc.dconv(samp_t)
            if _XSUB and _CUR[0] is not None and _XIDX in (-1, idx):          # dev: the hazard hunt's sub-layer serialisation points
                if _XSUB & 1:
                    _CUR[0].wait_stream(_CUR[1])          # main waits for the time stream's DConv
                if _XSUB & 2:
                    _CUR[1].wait_stream(_CUR[0])          # the time stream's stride-4 node waits for what main has enqueued
            if idx < Lt - 1:
                nxt_t = CL_DCONV and CL_TIME_DCONV and self.time_encoder[idx + 1].dconv.cl_ok(len_t // 4)
                et, samp_t = clchain.enc_mid(dt_, tenc.rewrite, self.time_encoder[idx + 1].conv, None, B, y_cl=nxt_t, fold=True)
            else:
                et, xt = clchain.enc_tail(dt_, tenc.rewrite, B)
                xt = xt.squeeze(2)
            saved_t.append(et)
        else:
            lengths_t.append(xt.shape[-1])
            if saved_t and saved_t[-1] is xt:          # xt is the previous layer's output = a skip connection: see fork
                xt, alias = tenc(xt, fork=True)
                if alias is not None:
                    saved_t[-1] = alias
            else:
                xt = tenc(xt)
            if not tenc.empty:
                saved_t.append(xt)
            else:
                inject = xt
        return xt, samp_t, len_t, inject

    def forward(self, input):
        if input.ndim != 3 or input.shape[1] != self.audio_channels:
            raise ValueError(f"expected (batch, {self.audio_channels}, frames), got {tuple(input.shape)}")
        ops._req(input, "input")
        B, Cin, length = input.shape
        hl = self.hop_length
        le = math.ceil(length / hl)
        pad = hl // 2 * 3
        if input.requires_grad:
            raise NotImplementedError("HDemucs: gradient w.r.t. the input waveform is not on the reference's path")
        saved, saved_t, lengths, lengths_t = [], [], [], []
        Lc = self._cl_layers(le, input.device)
        Lt = self._cl_layers_time(length, input.device, Lc)
        samp = samp_t = None
        len_t = length
        import contextlib
        # ON again since the GroupNorm statistics are stored per chunk (with epilogue statistics -- a zero fill followed by
        # fp64 atomics -- one clip of a batch came out 1e-3 wrong in ~3 % of forward passes whenever this stream ran beside the main one;
        # scripts/probes/batch_invariance_loop.py is the acceptance test: 0 of 200).
        two = TWO_STREAMS and input.is_cuda and Lt > 0 and not _data_parallel()
        xt, meant, stdt = nnops.row_standardize(input, 1e-5)        # over (C, T) per clip
        if two:
            main_s, time_s = torch.cuda.current_stream(), _time_stream(input.device)
            time_s.wait_stream(main_s)
            xt.record_stream(time_s)
        tctx = (lambda: torch.cuda.stream(time_s)) if two else contextlib.nullcontext
        _CUR[0], _CUR[1] = (main_s, time_s) if two else (None, None)
        Fq = self.nfft // 2
        # Frame-major ends (round 6): on the channels-last trunk the spectrum never takes torch.stft's [bin][frame] layout, which costs
        # the FFT kernels 8-byte pieces of 128-byte lines on both sides of the U-Net.  _spec stores frame-major (full lines); the first
        # convolution's im2col operand is built straight from it with the standardisation folded in (no separate affine pass); the
        # de-standardisation behind the last transposed convolution is fused with the layout change to frame-major, which _ispec reads.
        fm = (FM_ENDS and Lc > 0 and CL_ENDS and CL_DCONV and Cin == 1 and Fq % 4 == 0 and len(self.sources) == 1
              and self.freq_encoder[0].dconv.cl_ok())
        if fm:
            spec_fm = stft.stft(input.reshape(B * Cin, length), self.nfft, hl, mode="complex_fm", normalized=True,
                                bins=Fq, frame0=2, frames_out=le, extra_pad=(pad, pad + le * hl - length)).detach()    # (B, le, Fq, 2)
            mean, std, coef_a, coef_b = nnops.row_moments(spec_fm, 1e-5)                                             # over (C, Fr, T) per clip
            x = None
        else:
            # _spec + _magnitude: STFT straight into complex-as-channels (B, 2*Cin, nfft/2, le)
            cac = stft.stft(input.reshape(B * Cin, length), self.nfft, hl, mode="cac", normalized=True,
                            bins=self.nfft // 2, frame0=2, frames_out=le, extra_pad=(pad, pad + le * hl - length))
            x = cac.view(B, Cin, 2, Fq, le).reshape(B, Cin * 2, Fq, le)
            x, mean, std = nnops.row_standardize(x.detach(), 1e-5)      # over (C, Fr, T) per clip, unbiased std
        for idx, encode in enumerate(self.freq_encoder):
            lengths.append(le if x is None else x.shape[-1])
            inject = None
            if idx < len(self.time_encoder):
                if two and (_XSYNC & 2) and _XIDX in (-1, idx):
                    time_s.wait_stream(main_s)
                with tctx():
                    xt, samp_t, len_t, inject = self._time_encoder_step(idx, Lt, B, saved_t, lengths_t, xt, samp_t, len_t)
                if two and (_XSYNC & 1) and _XIDX in (-1, idx):
                    main_s.wait_stream(time_s)
                if two and inject is not None:                 # the merge: layer 4 of the frequency branch reads the time branch
                    main_s.wait_stream(time_s)
                    inject.record_stream(main_s)
            if idx < Lc:
                # channels-last trunk: the DConv branch on (B * Fr, C, T) samples, everything between two branches in one node
                dcl = CL_DCONV and encode.dconv.cl_ok()      # this layer's DConv branch runs on channels-last samples
                if idx == 0:
                    samp = clchain.head_conv_fm(spec_fm, coef_a, coef_b, encode.conv) if fm else encode.head(x, cl=dcl, ends=CL_ENDS)
                d = encode.dconv.forward_cl(samp) if dcl else encode.dconv(samp)
                if getattr(self, "_dbg", None) is not None and not dcl:        # dev: per-clip checksums around the channel-major DConv branches
                    csn = lambda t: t.float().abs().reshape(B, -1).sum(1) if t.is_contiguous() else t.float().abs().contiguous().reshape(B, -1).sum(1)
                    self._dbg.setdefault("samp", {})[idx] = csn(samp)
                    self._dbg.setdefault("d", {})[idx] = csn(d)
                if idx < Lc - 1:
                    emb_rows = self.freq_emb.table() * self.freq_emb_scale if (idx == 0 and self.freq_emb is not None) else None
                    nxt = CL_DCONV and self.freq_encoder[idx + 1].dconv.cl_ok()
                    e, samp = clchain.enc_mid(d, encode.rewrite, self.freq_encoder[idx + 1].conv, emb_rows, B, y_cl=nxt)
                else:
                    e, x = clchain.enc_tail(d, encode.rewrite, B)
                saved.append(e)
                continue
            if saved and saved[-1] is x:
                x, alias = encode(x, inject, fork=True)
                if alias is not None:
                    saved[-1] = alias
            else:
                x = encode(x, inject)
            if idx == 0 and self.freq_emb is not None:
                emb = self.freq_emb.table().t()[None, :, :, None]
                x = nnops.add(x, emb, self.freq_emb_scale)
            saved.append(x)
        x = ops.zeros(x.shape, x.device)
        xt = ops.zeros(x.shape, x.device)
        if getattr(self, "_dbg", None) is not None:        # dev (scripts/probes/first_wrong_tensor.py): per-clip checksums of the skips
            cs = lambda t: t.float().abs().sum(dim=tuple(range(1, t.dim()))) if t.shape[0] == B else t.float().abs().view(B, -1).sum(1)
            self._dbg["saved"] = [cs(t) for t in saved]
            self._dbg["x"] = cs(x)
            with tctx():
                self._dbg["saved_t"] = [cs(t) for t in saved_t]
        offset = self.depth - len(self.time_decoder)
        fadd = tadd = False                      # the previous layer already added this layer's skip (activation_add)
        for idx, decode in enumerate(self.freq_decoder):
            j = self.depth - 1 - idx
            if j < Lc:
                pre = None
                lengths.pop(-1)
                if j == Lc - 1:                   # layers Lc - 1 .. 0 in one node; the last transposed convolution (C -> 2 audio) stays channel-major
                    skips = [saved.pop(-1) for _ in range(Lc)]
                    last = self.freq_decoder[-1]
                    if CL_ENDS and last.last and last.conv_tr.out_channels <= 2:
                        x = clchain.freq_decoder(x, skips, list(self.freq_decoder[idx:]), tail=True)
                    else:
                        y0 = clchain.freq_decoder(x, skips, list(self.freq_decoder[idx:]))
                        full = (y0.shape[2] - 1) * last.stride + last.kernel_size
                        x = ops.conv_transpose2d(y0, last.conv_tr.weight, last.conv_tr.bias, (last.stride, 1), (1, 1), (last.pad, 0),
                                                 (full - 2 * last.pad, y0.shape[3]))
                        if not last.last:
                            x = nnops.gelu(x)
            else:
                skip = saved.pop(-1)
                x, pre = decode(x, skip, lengths.pop(-1), next_skip=saved[-1] if (saved and j != Lc) else None, skip_added=fadd)
                fadd = decode.fused_next_add
            if idx >= offset and two and (self.time_decoder[idx - offset].empty):
                time_s.wait_stream(main_s)                     # the empty time layer reads the frequency branch's layer-4 tensor
                pre.record_stream(time_s)
            if two and (_XSYNC & 8):
                time_s.wait_stream(main_s)
            with tctx():
                if idx >= offset and j < Lt:
                    length_t = lengths_t.pop(-1)
                    if j == Lt - 1:
                        skips_t = [saved_t.pop(-1) for _ in range(Lt)]
                        lastt = self.time_decoder[-1]
                        if CL_ENDS and lastt.last and lastt.conv_tr.out_channels <= 2:
                            xt = clchain.freq_decoder(xt.unsqueeze(2), skips_t, list(self.time_decoder[idx - offset:]), fold=True, tail=True).squeeze(2)
                        else:
                            yt0 = clchain.freq_decoder(xt.unsqueeze(2), skips_t, list(self.time_decoder[idx - offset:]), fold=True)
                            xt = ops.conv_transpose1d(yt0.squeeze(2), lastt.conv_tr.weight, lastt.conv_tr.bias, lastt.stride, 1, lastt.pad,
                                                      lengths_t[0] if lengths_t else length_t)
                            if not lastt.last:
                                xt = nnops.gelu(xt)
                elif idx >= offset:
                    tdec = self.time_decoder[idx - offset]
                    length_t = lengths_t.pop(-1)
                    if tdec.empty:
                        xt, _ = tdec(pre[:, :, 0], None, length_t)
                        tadd = False
                    else:
                        skip_t = saved_t.pop(-1)
                        xt, _ = tdec(xt, skip_t, length_t, next_skip=saved_t[-1] if saved_t else None, skip_added=tadd)
                        tadd = tdec.fused_next_add
            if two and (_XSYNC & 4):
                main_s.wait_stream(time_s)
        if two:
            main_s.wait_stream(time_s)
            xt.record_stream(main_s)
            if _XSYNC & 16:
                torch.cuda.synchronize()
        S = len(self.sources)
        if S != 1:
            raise NotImplementedError("multi-source de-standardisation")
        if fm and x.shape == (B, 2, Fq, le):
            # de-standardisation + _mask's layout change in one pass, then _ispec on the frame-major spectrum
            spec = nnops.cm_to_fm_affine(x, std, mean)                 # (B, le, Fq, 2)
            xo = stft.istft(spec, self.nfft, hl, mode="complex_fm", normalized=True, frames=le + 4, frame0=2, crop=pad,
                            length=length).view(B, S, Cin, length)
        else:
            x = nnops.row_affine(x.reshape(B, -1), std, mean)          # S == 1 for RemFX: one (std, mean) per clip
            # _mask + _ispec: (B, S, Cin*2, Fq, le) complex-as-channels -> time, one row per (b, s, c)
            spec = x.view(B * S * Cin, 2, Fq, le)
            xo = stft.istft(spec, self.nfft, hl, mode="cac", normalized=True, frames=le + 4, frame0=2, crop=pad,
                            length=length).view(B, S, Cin, length)
        xt = nnops.row_affine(xt.reshape(B, -1), stdt, meant).view(B, S, -1, length)
        return nnops.add(xt.reshape(B, S * Cin, 1, length), xo.reshape(B, S * Cin, 1, length)).view(B, S, Cin, length)
