"""CPU oracle (TEST INFRASTRUCTURE ONLY): asteroid.models.DPTNet as RemFX configures it
(reference remfx/models.py:327-344 `DPTNetModel`, cfg/model/dptnet.yaml: n_src 1, in_chan 64, out_chan 64,
chunk_size 100, n_repeats 2, fb_name "free", kernel_size 16, n_filters 64, stride 8).

PARITY UNPINNED: `asteroid` is a bare dependency in the reference's setup.py and is in neither /root/reference nor this
image.  This is a plain-torch fp32 restatement of the published architecture (asteroid/models/dptnet.py,
asteroid/masknn/attention.py `DPTransformer` + `ImprovedTransformedLayer`, asteroid/masknn/norms.py `GlobLN`,
asteroid/dsp/overlap_add.py `DualPathProcessing`, asteroid_filterbanks `FreeFB` / `Encoder` / `Decoder`), with
asteroid's state_dict names, so that a released checkpoint would load:
    encoder.filterbank._filters (64, 1, 16)         decoder.filterbank._filters (64, 1, 16)
    masker.in_norm.{gamma,beta}
    masker.layers.{r}.{0 intra,1 inter}.mha.{in_proj_weight,in_proj_bias,out_proj.weight,out_proj.bias}
    masker.layers.{r}.{i}.recurrent.{weight,bias}_{ih,hh}_l0[_reverse]     masker.layers.{r}.{i}.linear.{weight,bias}
    masker.layers.{r}.{i}.norm_mha.{gamma,beta}     masker.layers.{r}.{i}.norm_ff.{gamma,beta}
    masker.first_out.0.weight (PReLU)  masker.first_out.1.{weight,bias} (Conv2d 1x1)
    masker.net_out.0.{weight,bias}     masker.net_gate.0.{weight,bias}
Algorithm: x (B, T) -> relu(conv1d(x, filters, stride 8)) = tf (B, 64, F); gLN; chunks of 100 frames at hop 50 (zero padding
of one chunk either side); n_repeats x [intra-chunk layer over the 100 frames of every chunk, inter-chunk layer over the
chunks at every intra position]; layer = x + MHA(x) -> gLN -> x + Linear(relu(BiLSTM_256(x))) -> gLN; PReLU -> 1x1 conv ->
overlap-add / 2 -> tanh(conv) * sigmoid(conv) -> relu = mask; tf * mask -> conv_transpose1d(filters, stride 8) -> pad / crop
to T.  Output (B, n_src, T).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

EPS = 1e-8


class GlobLN(nn.Module):
    def __init__(self, channel_size):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(channel_size))
        self.beta = nn.Parameter(torch.zeros(channel_size))

    def forward(self, x):
        dims = list(range(1, x.dim()))
        mean = x.mean(dim=dims, keepdim=True)
        var = torch.pow(x - mean, 2).mean(dim=dims, keepdim=True)
        normed = (x - mean) / (var + EPS).sqrt()
        return (normed.transpose(1, -1) * self.gamma + self.beta).transpose(1, -1)


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


class ImprovedTransformedLayer(nn.Module):
    def __init__(self, embed_dim, n_heads, dim_ff, dropout=0.0, bidirectional=True):
        super().__init__()
        self.mha = nn.MultiheadAttention(embed_dim, n_heads, dropout=dropout)
        self.recurrent = nn.LSTM(embed_dim, dim_ff, bidirectional=bidirectional, batch_first=True)
        self.linear = nn.Linear(2 * dim_ff if bidirectional else dim_ff, embed_dim)
        self.norm_mha = GlobLN(embed_dim)
        self.norm_ff = GlobLN(embed_dim)

    def forward(self, x):                                   # (batch, channels, seq)
        tomha = x.permute(2, 0, 1)
        out = self.mha(tomha, tomha, tomha)[0]
        x = out.permute(1, 2, 0) + x
        x = self.norm_mha(x)
        out = self.linear(F.relu(self.recurrent(x.transpose(1, -1))[0]))
        x = out.transpose(1, -1) + x
        return self.norm_ff(x)


class DPTransformer(nn.Module):
    def __init__(self, in_chan, n_src, n_heads=4, ff_hid=256, chunk_size=100, hop_size=None, n_repeats=6):
        super().__init__()
        assert in_chan % n_heads == 0
        self.in_chan, self.n_src, self.chunk_size = in_chan, n_src, chunk_size
        self.hop_size = hop_size if hop_size is not None else chunk_size // 2
        self.in_norm = GlobLN(in_chan)
        self.layers = nn.ModuleList([nn.ModuleList([ImprovedTransformedLayer(in_chan, n_heads, ff_hid),
                                                    ImprovedTransformedLayer(in_chan, n_heads, ff_hid)])
                                     for _ in range(n_repeats)])
        self.first_out = nn.Sequential(nn.PReLU(), nn.Conv2d(in_chan, n_src * in_chan, 1))
        self.net_out = nn.Sequential(nn.Conv1d(in_chan, in_chan, 1), nn.Tanh())
        self.net_gate = nn.Sequential(nn.Conv1d(in_chan, in_chan, 1), nn.Sigmoid())

    def forward(self, w):
        w = self.in_norm(w)
        frames = w.shape[-1]
        batch, chan, _ = w.shape
        u = F.unfold(w.unsqueeze(-1), kernel_size=(self.chunk_size, 1), padding=(self.chunk_size, 0), stride=(self.hop_size, 1))
        u = u.reshape(batch, chan, self.chunk_size, -1)
        n_chunks = u.shape[-1]
        for intra, inter in self.layers:
            v = u.transpose(1, -1).reshape(batch * n_chunks, self.chunk_size, chan).transpose(1, -1)
            v = intra(v)
            u = v.reshape(batch, n_chunks, chan, self.chunk_size).transpose(1, -1).transpose(1, 2)
            v = u.transpose(1, 2).reshape(batch * self.chunk_size, chan, n_chunks)
            v = inter(v)
            u = v.reshape(batch, self.chunk_size, chan, n_chunks).transpose(1, 2)
        out = self.first_out(u)
        out = out.reshape(batch * self.n_src, self.in_chan, self.chunk_size, n_chunks)
        to_fold = out.reshape(batch * self.n_src, self.in_chan * self.chunk_size, n_chunks)
        out = F.fold(to_fold, (frames, 1), kernel_size=(self.chunk_size, 1), padding=(self.chunk_size, 0), stride=(self.hop_size, 1))
        out = out / (float(self.chunk_size) / self.hop_size)
        out = out.reshape(batch * self.n_src, self.in_chan, frames)
        out = self.net_out(out) * self.net_gate(out)
        return F.relu(out.reshape(batch, self.n_src, self.in_chan, frames))


class DPTNet(nn.Module):
    def __init__(self, n_src, n_heads=4, ff_hid=256, chunk_size=100, hop_size=None, n_repeats=6, norm_type="gLN",
                 ff_activation="relu", encoder_activation="relu", mask_act="relu", bidirectional=True, dropout=0, in_chan=None,
                 fb_name="free", kernel_size=16, n_filters=64, stride=8, sample_rate=8000, **fb_kwargs):
        super().__init__()
        assert fb_name == "free" and norm_type == "gLN" and bidirectional and dropout == 0
        assert in_chan is None or in_chan == n_filters
        self.n_src, self.stride = n_src, stride
        self.encoder = _Coder(FreeFB(n_filters, kernel_size, stride))
        self.decoder = _Coder(FreeFB(n_filters, kernel_size, stride))
        self.masker = DPTransformer(n_filters, n_src, n_heads, ff_hid, chunk_size, hop_size, n_repeats)

    def forward(self, wav):                                  # (B, T) -> (B, n_src, T)
        x = wav.unsqueeze(1)
        tf = F.relu(F.conv1d(x, self.encoder.filterbank._filters, stride=self.stride))
        masks = self.masker(tf)
        masked = masks * tf.unsqueeze(1)
        B, S, N, Fr = masked.shape
        dec = F.conv_transpose1d(masked.reshape(B * S, N, Fr), self.decoder.filterbank._filters, stride=self.stride)
        dec = dec.reshape(B, S, -1)
        T = wav.shape[-1]
        return F.pad(dec, (0, T - dec.shape[-1])) if dec.shape[-1] < T else dec[..., :T]
