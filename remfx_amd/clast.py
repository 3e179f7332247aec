"""Channels-last bf16 trunk of the frequency branch of Hybrid Demucs (bf16 arithmetic mode = BASELINE config 3,
`trainer.precision=bf16-mixed`; torchaudio HDemucs `freq_encoder` / `freq_decoder` behind remfx/models.py:308,317).

Tensors here are torch.bfloat16 of shape (N, A, B, C), contiguous, C % 8 == 0: the channels of one position are contiguous, so an
MFMA operand fragment is one 16-byte group and the kernels (csrc/cl_conv.hip, cl_wgrad.hip, cl_elem.hip) move operands global ->
LDS by DMA.  This module holds the host side: the packing index of every layer form (which weight element lands in which MFMA
fragment cell), descriptors, and the autograd nodes of the encoder / decoder chains.
"""
import ctypes as C
import os as _os

import numpy as np
import torch

from . import _lib, ops
from ._lib import ClConvDesc, ClTensor, ClWgradDesc, check

EPI = {"store": 0, "gelu": 1, "glu": 2, "dgelu": 3, "dglu": 4, "store_cm": 5}


def _stream():
    return C.c_void_p(ops.raw_stream())


def cl_tensor(t, c0=0):
    """rfx_cl_tensor view of a (N, A, B, C) bf16 tensor whose last two axes are dense (stride(3) == 1, stride(2) == C_stored)."""
    ct = ClTensor()
    if t is None:
        return ct
    if t.dtype != torch.bfloat16 or t.dim() != 4 or t.stride(3) != 1 or not t.is_cuda:
        raise ValueError(f"channels-last operand must be a 4-D bf16 GPU tensor with dense channels (got {t.dtype}, {tuple(t.shape)}, {t.stride()})")
    ct.p = t.data_ptr()
    ct.ns, ct.as_, ct.bs, ct.c0 = t.stride(0), t.stride(1), t.stride(2), c0
    return ct


def empty(N, A, B, Cc, device):
    return torch.empty((N, A, B, Cc), device=device, dtype=torch.bfloat16)


FROM_CM_MODE = {"store": 0, "dgelu": 1, "dglu": 2, "gelu": 3}


def from_cm(x, out=None, res=None, aux=None, mode="store"):
    """(N, C, A, B) fp32 / bf16 channel-major (any strides with B contiguous) -> (N, A, B, C) bf16 channels-last, fused with what
    would follow: v = x (+ res); "store": v; "dgelu": v * gelu'(aux); "dglu": GLU backward against aux = stored [a | b] (out then
    has 2 C channels)."""
    N, Cc, A, B = x.shape
    if x.stride(3) != 1:
        x = x.contiguous()
    if out is None:
        out = empty(N, A, B, 2 * Cc if mode == "dglu" else Cc, x.device)
    ct, cr, ca = cl_tensor(out), cl_tensor(res), cl_tensor(aux)
    check(_lib.lib().rfx_cl_from_cm(C.c_void_p(x.data_ptr()), int(x.dtype == torch.bfloat16), x.stride(0), x.stride(1), x.stride(2),
                                    N, Cc, A, B, C.byref(ct), C.byref(cr), C.byref(ca), FROM_CM_MODE[mode], _stream()), "rfx_cl_from_cm")
    return out


def to_cm(x, dtype=torch.float32, out=None, aux16=None):
    """(N, A, B, C) bf16 channels-last -> (N, C, A, B) channel-major fp32 / bf16; aux16 (bf16 tensor with out's strides):
    out = x * gelu'(aux16)."""
    N, A, B, Cc = x.shape
    if out is None:
        out = torch.empty((N, Cc, A, B), device=x.device, dtype=dtype)
    ct = cl_tensor(x)
    check(_lib.lib().rfx_cl_to_cm(C.byref(ct), N, Cc, A, B, C.c_void_p(out.data_ptr()), int(out.dtype == torch.bfloat16),
                                  out.stride(0), out.stride(1), out.stride(2),
                                  C.c_void_p(aux16.data_ptr()) if aux16 is not None else None, _stream()), "rfx_cl_to_cm")
    return out


def dgelu(g, z):
    """g * gelu'(z) for dense channels-last tensors of one shape."""
    if g.shape != z.shape or not g.is_contiguous() or not z.is_contiguous():
        raise ValueError("dgelu: dense tensors of one shape")
    out = torch.empty_like(g)
    check(_lib.lib().rfx_cl_dgelu(C.c_void_p(g.data_ptr()), C.c_void_p(z.data_ptr()), C.c_void_p(out.data_ptr()), g.numel(), _stream()),
          "rfx_cl_dgelu")
    return out


def dglu(g, zab):
    """GLU backward: g (..., C) against the stored zab (..., 2C) = [a | b], dense channels-last tensors -> (..., 2C)."""
    Cc = g.shape[-1]
    if zab.shape[-1] != 2 * Cc or g.numel() * 2 != zab.numel() or not g.is_contiguous() or not zab.is_contiguous():
        raise ValueError("dglu: dense tensors (..., C) and (..., 2C)")
    out = torch.empty_like(zab)
    check(_lib.lib().rfx_cl_dglu(C.c_void_p(g.data_ptr()), C.c_void_p(zab.data_ptr()), C.c_void_p(out.data_ptr()), g.numel() // Cc, Cc,
                                 _stream()), "rfx_cl_dglu")
    return out


# ---- tile choice ----------------------------------------------------------------------------------------------------------------
BM96_NTC = _os.environ.get("RFX_CL_BM96_NTC", "0") != "0"   # 96-row tiles for every multi-column-tap layer (the first form of round 5)
BM96_MAX_K = int(_os.environ.get("RFX_CL_BM96_K", "0"))    # 96-row tiles for layers whose reduction is at most this long (0: A/B off)


def pick_bm(M, K=1 << 30, NTC=1):
    """Rows per workgroup: 192 (two wave rows of 96) wherever M allows, else the smallest of 32 / 64 / 96 that holds M in the fewest
    tiles.  History (same-box A/B at 64 clips, r05): while the kernel carried every epilogue mode it needed 256 registers on the
    192-row tiles -- one workgroup per CU -- and 96-row tiles (two per CU: one's store tail under the other's loads) won for K <= 800
    and for the multi-column-tap layers (-5.5 ms on the step).  Compiled per mode the 192-row tiles fit 128 registers, run two per CU
    as well, and fetch every input slab once per 192 rows instead of once per 96: 107.6 -> 106.0 ms (RFX_CL_BM96_K=800 RFX_CL_BM96_NTC=1 restores the old rule)."""
    if M > 96:
        if ((BM96_MAX_K and K <= BM96_MAX_K) or (BM96_NTC and NTC > 1)) and M % 96 == 0:
            return 96
        return 192 if (M % 192 == 0 or M > 288) else 96
    if M > 64:
        return 96
    if M > 32:
        return 64
    return 32


class ConvForm:
    """One GEMM form of a layer: geometry of rfx_cl_conv + the gather index that packs its weight tensor.

    widx(m, r, t, ch) -> flat element index into the weight tensor (numpy int64 arrays broadcast together), -1 where the GEMM cell is
    structurally zero."""

    def __init__(self, M, Cin, NTR, NTC, da0, da_step, db0, db_step, SA, widx, G=1, g_off=0, Co=0, KS=None):
        self.M, self.Cin, self.NTR, self.NTC = M, Cin, NTR, NTC
        self.da0, self.da_step, self.db0, self.db_step, self.SA = da0, da_step, db0, db_step, SA
        self.G, self.g_off, self.Co = G, g_off, (Co or M)
        self.BM = pick_bm(M, NTR * NTC * Cin, NTC)
        self.MG = -(-M // self.BM)
        if KS is None:
            KS = 1 if NTC > 1 else (2 if Cin % 32 == 0 else 1)
        self.KS = KS
        if Cin % (16 * KS):
            raise ValueError(f"channels-last GEMM: {Cin} input channels are not a multiple of {16 * KS}")
        self.NCH = Cin // (16 * KS)
        self.idx = self._build_index(widx)
        self._dev = {}

    def _build_index(self, widx):
        MT = self.BM // 32
        U = self.NTR * self.NCH
        shape = (self.NTR, self.NCH, self.MG, self.NTC, self.KS, MT, 64, 8)
        r, c, mg, t, ks, mt, lane, e = np.meshgrid(*[np.arange(s, dtype=np.int64) for s in shape], indexing="ij", sparse=True)
        m = mg * self.BM + mt * 32 + (lane & 31)
        ch = c * (16 * self.KS) + ks * 16 + 8 * (lane >> 5) + e
        ok = (m < self.M) & (ch < self.Cin)
        m_c = np.minimum(m, self.M - 1)
        ch_c = np.minimum(ch, self.Cin - 1)
        idx = np.broadcast_to(widx(m_c, r, t, ch_c), shape)
        idx = np.where(np.broadcast_to(ok, shape), idx, -1)
        assert idx.size == U * self.MG * self.NTC * self.KS * MT * 512
        return np.ascontiguousarray(idx.reshape(-1).astype(np.int32))


def _dev_index(form, device):
    """The form's gather index on `device` (uploaded once; kept on the form object, whose lifetime it shares)."""
    key = str(device)
    v = form._dev.get(key)
    if v is None:
        v = torch.from_numpy(form.idx).to(device)
        form._dev[key] = v
    return v


def pack(form, w):
    """Packed MFMA A fragments (bf16) of weight tensor w for one GEMM form: one gather launch."""
    idx = _dev_index(form, w.device)
    wf = w if w.is_contiguous() else w.contiguous()
    out = torch.empty(idx.numel(), device=w.device, dtype=torch.bfloat16)
    check(_lib.lib().rfx_cl_pack(C.c_void_p(wf.data_ptr()), C.c_void_p(idx.data_ptr()), idx.numel(), C.c_void_p(out.data_ptr()),
                                 _stream()), "rfx_cl_pack")
    return out


def conv(form, apack, x, N, IA, IB, OA, mode, bias=None, out0=None, out1=None, aux0=None, res=None, OAo=0, x_c0=0, wrapb=False,
         rowadd=None, cm_out=None, cm_fold=False):
    """Launch rfx_cl_conv for `form` on the channels-last operand x; outputs / auxiliaries are channels-last tensors."""
    d = ClConvDesc()
    d.inp = cl_tensor(x, x_c0)
    d.N, d.IA, d.IB, d.OA, d.OB, d.SA = N, IA, IB, OA, IB, form.SA
    d.NTR, d.NCH, d.NTC, d.KS = form.NTR, form.NCH, form.NTC, form.KS
    d.da0, d.da_step, d.db0, d.db_step = form.da0, form.da_step, form.db0, form.db_step
    d.wrapb = int(wrapb)
    d.apack = apack.data_ptr()
    d.M, d.BM, d.mode = form.M, form.BM, EPI[mode]
    d.G, d.g_off, d.OAo, d.Co = form.G, form.g_off, OAo, form.Co
    d.bias = bias.data_ptr() if bias is not None else None
    d.rowadd = rowadd.data_ptr() if rowadd is not None else None
    d.out0, d.out1, d.aux0, d.res = cl_tensor(out0), cl_tensor(out1), cl_tensor(aux0), cl_tensor(res)
    if cm_out is not None:                       # (N, Co, rows, positions) fp32, positions contiguous
        if cm_out.dtype != torch.float32 or cm_out.stride(3) != 1:
            raise ValueError("store_cm: fp32 (N, C, rows, positions) with contiguous positions")
        d.cm_out, d.cm_ns, d.cm_cs, d.cm_as, d.cm_fold = cm_out.data_ptr(), cm_out.stride(0), cm_out.stride(1), cm_out.stride(2), int(cm_fold)
    check(_lib.lib().rfx_cl_conv(C.byref(d), _stream()), "rfx_cl_conv")


# ---- GEMM forms of the Hybrid Demucs layers ---------------------------------------------------------------------------------------
def form_conv_glu(Cout2, Cin, KA, KB):
    """Conv2d(Cin -> Cout2, (KA, KB), padding same) + GLU: GEMM rows interleaved (a_c, b_c); weight (Cout2, Cin, KA, KB)."""
    Ch = Cout2 // 2

    def widx(m, r, t, ch):
        oc = (m & 1) * Ch + (m >> 1)
        return ((oc * Cin + ch) * KA + r) * KB + t
    return ConvForm(Cout2, Cin, KA, KB, -(KA // 2), 1, -(KB // 2), 1, 1, widx)


def form_conv_dgrad(Cout, Cin, KA, KB):
    """Input gradient of Conv2d(Cin -> Cout, (KA, KB), stride 1, padding same): rows = Cin, reduction over Cout; weight (Cout, Cin, KA, KB)."""
    def widx(m, r, t, ch):
        return ((ch * Cin + m) * KA + (KA - 1 - r)) * KB + (KB - 1 - t)
    return ConvForm(Cin, Cout, KA, KB, -(KA // 2), 1, -(KB // 2), 1, 1, widx)


def form_conv(Cout, Cin, KA, KB):
    """Conv2d(Cin -> Cout, (KA, KB), stride 1, padding same), rows in natural order."""
    def widx(m, r, t, ch):
        return ((m * Cin + ch) * KA + r) * KB + t
    return ConvForm(Cout, Cin, KA, KB, -(KA // 2), 1, -(KB // 2), 1, 1, widx)


def form_convtr_s4(Cin, Cout):
    """ConvTranspose2d(Cin -> Cout, (8, 1), stride (4, 1)) cropped by 2 rows on each side, as ONE GEMM over 2 row taps: super-row q
    (0 .. IA) produces output rows 4 q + psi - 2, psi = 0..3, from input rows q - 1 (kernel tap psi + 4) and q (tap psi).
    Weight (Cin, Cout, 8, 1).  Rows m = psi * Cout + co."""
    def widx(m, r, t, ch):
        psi, co = m // Cout, m % Cout
        k = psi + 4 * (1 - r)                      # r = 0: input row q - 1; r = 1: input row q
        return (ch * Cout + co) * 8 + k
    return ConvForm(4 * Cout, Cin, 2, 1, -1, 1, 0, 0, 1, widx, G=4, g_off=-2, Co=Cout)


def form_conv_s4(Cout, Cin):
    """Conv2d(Cin -> Cout, (8, 1), stride (4, 1), padding (2, 0)): 8 row taps, input row 4 oa + k - 2.  Weight (Cout, Cin, 8, 1).
    Also the input gradient of form_convtr_s4's layer when called with its weight transposed (see form_convtr_s4_dgrad)."""
    def widx(m, r, t, ch):
        return (m * Cin + ch) * 8 + r
    return ConvForm(Cout, Cin, 8, 1, -2, 1, 0, 0, 4, widx)


def form_convtr_s4_dgrad(Cin, Cout):
    """Input gradient of the ConvTranspose2d above: dy[ia][ci] = sum_k sum_co w[ci][co][k] dz[4 ia + k - 2][co]."""
    def widx(m, r, t, ch):
        return (m * Cout + ch) * 8 + r
    return ConvForm(Cin, Cout, 8, 1, -2, 1, 0, 0, 4, widx)


def form_conv_s4_dgrad(Cout, Cin):
    """Input gradient of Conv2d(Cin -> Cout, (8, 1), stride (4, 1), padding (2, 0)) as a merged 2-row-tap GEMM (the transposed
    convolution): rows m = rho * Cin + ci, output row 4 q + rho - 2 from gradient rows q - 1 (tap rho + 4) and q (tap rho)."""
    def widx(m, r, t, ch):
        rho, ci = m // Cin, m % Cin
        k = rho + 4 * (1 - r)
        return (ch * Cin + ci) * 8 + k
    return ConvForm(4 * Cin, Cout, 2, 1, -1, 1, 0, 0, 1, widx, G=4, g_off=-2, Co=Cin)


# ---- stride-4 convolutions ALONG the position axis (the time branch: A = 1): the operand is read / written through its folded view
# [L / 4][4 C] -- four consecutive positions are 4 C consecutive channels of one folded position -- which turns the (8, stride 4)
# kernel into a 3-tap stride-1 convolution over folded positions whose weight image has structural zeros (8 of the 12 (tap,
# sub-position) slots are taps of the kernel); the index functions return -1 there.
def _k_down(t, j):
    """kernel tap read by folded tap t (offset t - 1), sub-position j of the FINE operand, for a coarse position: 4 (t - 1) + j + 2"""
    return 4 * (t - 1) + j + 2


def _k_up(t, j):
    """kernel tap that sends coarse position q' + t - 1 to fine position 4 q' + j: j + 6 - 4 t"""
    return j + 6 - 4 * t


def form_conv_s4_fold(Cout, Cin):
    """Conv1d(Cin -> Cout, 8, stride 4, padding 2) over folded input (.., L / 4, 4 Cin); weight (Cout, Cin, 8)."""
    def widx(m, r, t, ch):
        j, ci = ch // Cin, ch % Cin
        k = _k_down(t, j)
        return np.where((k >= 0) & (k < 8), (m * Cin + ci) * 8 + np.clip(k, 0, 7), -1)
    return ConvForm(Cout, 4 * Cin, 1, 3, 0, 0, -1, 1, 1, widx)


def form_convtr_fold(Cin, Cout):
    """ConvTranspose1d(Cin -> Cout, 8, stride 4) cropped by 2 either side, output through its folded view (.., L, 4 Cout): rows
    m = j * Cout + co; weight (Cin, Cout, 8); bias index m % Cout."""
    def widx(m, r, t, ch):
        j, co = m // Cout, m % Cout
        k = _k_up(t, j)
        return np.where((k >= 0) & (k < 8), (ch * Cout + co) * 8 + np.clip(k, 0, 7), -1)
    return ConvForm(4 * Cout, Cin, 1, 3, 0, 0, -1, 1, 1, widx, Co=Cout)


def form_conv_s4_fold_dgrad(Cout, Cin):
    """Input gradient of form_conv_s4_fold's layer, written through the folded view (.., L / 4, 4 Cin): rows m = j * Cin + ci."""
    def widx(m, r, t, ch):
        j, ci = m // Cin, m % Cin
        k = _k_up(t, j)
        return np.where((k >= 0) & (k < 8), (ch * Cin + ci) * 8 + np.clip(k, 0, 7), -1)
    return ConvForm(4 * Cin, Cout, 1, 3, 0, 0, -1, 1, 1, widx, Co=Cin)


def form_convtr_fold_dgrad(Cin, Cout):
    """Input gradient of form_convtr_fold's layer: rows = Cin, reads the output gradient through its folded view (.., L, 4 Cout)."""
    def widx(m, r, t, ch):
        j, co = ch // Cout, ch % Cout
        k = _k_down(t, j)
        return np.where((k >= 0) & (k < 8), (m * Cout + co) * 8 + np.clip(k, 0, 7), -1)
    return ConvForm(Cin, 4 * Cout, 1, 3, 0, 0, -1, 1, 1, widx)


# ---- the network's ends: 1 - 2 channel tensors (the spectrogram's real / imaginary parts, the waveform) are no channels-last tensors.
# Their (8, stride 4) convolutions become 16-channel GEMMs over an im2col operand (rfx_cl_im2col_s4: channel k * Cs + c).
def im2col_s4(x, OA, OB, along_b):
    """x (N, Cs, IA, IB) fp32 channel-major (Cs = 1 | 2) -> (N, OA, OB, 16) bf16."""
    N, Cs, IA, IB = x.shape
    if x.dtype != torch.float32 or x.stride(3) != 1:
        x = x.float().contiguous()
    out = empty(N, OA, OB, 16, x.device)
    check(_lib.lib().rfx_cl_im2col_s4(C.c_void_p(x.data_ptr()), x.stride(0), x.stride(1), x.stride(2), N, Cs, IA, IB, OA, OB, int(along_b),
                                      C.c_void_p(out.data_ptr()), _stream()), "rfx_cl_im2col_s4")
    return out


def im2col_fm(spec, a, b):
    """spec (N, F, bins, 2) fp32 frame-major spectrum, a / b (N,) standardisation coefficients -> (N, bins / 4, F, 16) bf16: the im2col
    operand of the first convolution of the standardised spectrum (rfx_cl_im2col_fm)."""
    N, F, bins, two = spec.shape
    if two != 2 or spec.dtype != torch.float32 or not spec.is_contiguous() or bins % 4:
        raise ValueError("im2col_fm: contiguous (N, frames, bins, 2) fp32 with bins % 4 == 0")
    out = empty(N, bins // 4, F, 16, spec.device)
    check(_lib.lib().rfx_cl_im2col_fm(C.c_void_p(spec.data_ptr()), C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), N, F, bins,
                                      C.c_void_p(out.data_ptr()), _stream()), "rfx_cl_im2col_fm")
    return out


def form_head(Cout, Cs):
    """Conv (Cs -> Cout, 8 taps, stride 4, padding 2) on the im2col operand; weight (Cout, Cs, 8[, 1])."""
    def widx(m, r, t, ch):
        k, c = ch // Cs, ch % Cs
        return np.where(k < 8, (m * Cs + c) * 8 + np.minimum(k, 7), -1)
    return ConvForm(Cout, 16, 1, 1, 0, 0, 0, 0, 1, widx, KS=1)


def wform_head(Cout, Cs):
    """its weight gradient: P = output gradient (Cout), Q = the im2col operand."""
    def widx(m, r, t, c):
        return (m * Cs + c % Cs) * 8 + c // Cs
    return WgradForm(Cout, 8 * Cs, 1, 1, 1, 0, 0, 0, widx, Cout * Cs * 8)


def form_tail_tr(Cc, Cs):
    """ConvTranspose2d(Cc -> Cs, (8, 1), stride (4, 1)) cropped by 2 rows, merged row phases, stored channel-major (store_cm)."""
    def widx(m, r, t, ch):
        psi, co = m // Cs, m % Cs
        return (ch * Cs + co) * 8 + psi + 4 * (1 - r)
    return ConvForm(4 * Cs, Cc, 2, 1, -1, 1, 0, 0, 1, widx, G=4, g_off=-2, Co=Cs)


def form_tail_dgrad(Cc, Cs):
    """Input gradient of the last transposed convolution from the im2col of its output gradient: rows = Cc; weight (Cc, Cs, 8[, 1])."""
    def widx(m, r, t, ch):
        k, co = ch // Cs, ch % Cs
        return np.where(k < 8, (m * Cs + co) * 8 + np.minimum(k, 7), -1)
    return ConvForm(Cc, 16, 1, 1, 0, 0, 0, 0, 1, widx, KS=1)


def wform_tail(Cc, Cs):
    """its weight gradient: P = the layer's input (Cc), Q = the im2col of the output gradient."""
    def widx(m, r, t, c):
        return (m * Cs + c % Cs) * 8 + c // Cs
    return WgradForm(Cc, 8 * Cs, 1, 1, 1, 0, 0, 0, widx, Cc * Cs * 8, bias=False)


# ---- weight gradients -------------------------------------------------------------------------------------------------------------
CLW_SPLITS = int(_os.environ.get("RFX_CLW_SPLITS", "256"))      # workgroups of a channels-last weight gradient launch (A/B)


class WgradForm:
    """One weight-gradient GEMM (csrc/cl_wgrad.hip): D[m][(r, t, c)] = sum_pos P[pos][m] * Q[pos shifted by tap (r, t)][c].

    widx(m, r, t, c) -> flat index of that cell in the layer's weight tensor.  Chooses the tiling (rows per D tile, Q-channel slice,
    wave arrangement, prefetch depth) and builds the index map the fixed-order reduction scatters through."""

    LDS_MAX = 160 * 1024

    def __init__(self, M, Cq, NTR, NTC, SA, da0, db0, db_step, widx, wn, bias=True):
        self.M, self.Cq, self.NTR, self.NTC, self.SA, self.da0, self.db0, self.db_step = M, Cq, NTR, NTC, SA, da0, db0, db_step
        self.wn, self.bias = wn, bias
        self.RW = 3 if M > 64 else 2
        T = NTR * NTC
        nb = 1 if bias else 0
        best = None
        for CW in (96, 80, 64, 48, 32, 16):
            if CW > -(-Cq // 16) * 16:
                continue
            tiles = -(-(T * CW // 16) // 2) + nb
            if tiles > 16:
                continue
            ctn = -(-Cq // CW)
            cost = (ctn, ctn * CW - Cq)
            if best is None or cost < best[0]:
                best = (cost, CW, tiles)
        if best is None:
            raise ValueError(f"channels-last weight gradient: no tiling for {T} taps x {Cq} channels")
        _, self.CW, tiles = best
        self.WK = 1 if tiles > 8 else (2 if tiles > 4 else 4)
        self.WC = 8 // self.WK
        self.MTn, self.CTn = -(-M // (32 * self.RW)), -(-Cq // self.CW)
        self.DT = self.MTn * self.CTn
        hb = max(abs(db0 + t * db_step) for t in range(NTC))
        # 128 positions per step (twice the MFMA work per barrier and per scalar bookkeeping section) when two steps of prefetch
        # still fit the LDS, else 64; the prefetch depth itself measured irrelevant beyond 1 (r05 ablations)
        self.ahead = None
        for PW, amin in (((128, 2), (64, 1)) if _os.environ.get("RFX_CLW_PW", "128") == "128" else ((64, 1),)):
            NP = PW // 16 * self.RW
            NQ = -(-((PW + 2 * hb) * self.CW * 2) // 1024)
            PPW = -(-(NP + SA * NQ) // 8)
            if PPW > 6 or (PW // 16) % self.WK:
                continue
            for ahead in (4, 3, 2, 1):
                R = -(-(NTR + ahead * SA) // SA) * SA
                if ahead >= amin and (ahead + 1) * NP * 1024 + R * NQ * 1024 <= self.LDS_MAX and (ahead - 1) * PPW <= 24:
                    self.ahead, self.PW = ahead, PW
                    break
            if self.ahead is not None:
                break
        if self.ahead is None:
            raise ValueError("channels-last weight gradient: tile does not fit LDS")
        self.map = self._build_map(widx)
        self._dev = {}

    def _build_map(self, widx):
        RW, NT, WC = self.RW, 2, self.WC
        NCG = self.CW // 16
        HN = self.NTR * self.NTC * NCG
        bias_tile = -(-HN // 2)
        shape = (self.MTn, self.CTn, WC, RW, NT, 16, 64)
        mt, ct, cw, i, t, r, lane = np.meshgrid(*[np.arange(s, dtype=np.int64) for s in shape], indexing="ij", sparse=True)
        m = mt * 32 * RW + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
        l31 = lane & 31
        tile = cw * NT + t
        hh = 2 * tile + (l31 >> 4)
        cgl, tt = hh % NCG, hh // NCG
        tc, tr = tt % self.NTC, tt // self.NTC
        c = ct * self.CW + cgl * 16 + (l31 & 15)
        ok = (hh < HN) & (m < self.M) & (c < self.Cq)
        idx = widx(np.minimum(m, self.M - 1), np.minimum(tr, self.NTR - 1), tc, np.minimum(c, self.Cq - 1))
        out = np.where(np.broadcast_to(ok, shape), np.broadcast_to(idx, shape), -1)
        if self.bias:
            isb = (tile == bias_tile) & (l31 == 0) & (ct == 0) & (m < self.M)
            out = np.where(np.broadcast_to(isb, shape), np.broadcast_to(self.wn + m, shape), out)
        flat = out.reshape(-1)
        used = flat[flat >= 0]
        assert used.size == np.unique(used).size, "weight-gradient map: a destination is written twice"
        return np.ascontiguousarray(flat.astype(np.int32))

    def splits(self, steps):
        """Position splits: one workgroup per CU in all (a workgroup fills a CU's LDS), every split non-empty."""
        S = max(1, min(steps, -(-CLW_SPLITS // self.DT)))
        sps = -(-steps // S)
        return -(-steps // sps)


def _dev_map(form, device):
    key = str(device)
    v = form._dev.get(key)
    if v is None:
        v = torch.from_numpy(form.map).to(device)
        form._dev[key] = v
    return v


def wgrad(form, p, q, N, OA, IA, B, dw, db=None, accumulate=False, p_c0=0, q_c0=0):
    """dw (flat fp32 view of the weight gradient, form.wn elements) and db (fp32, form.M) from the channels-last operands p
    (N, OA, B, >= M channels) and q (N, IA, B, >= Cq channels).  Two launches: partial sums per position split, then the fixed-order
    reduction (deterministic: no atomics)."""
    if form.bias and db is None:
        raise ValueError("this weight-gradient form carries the bias gradient: pass db")
    d = ClWgradDesc()
    d.p, d.q = cl_tensor(p, p_c0), cl_tensor(q, q_c0)
    d.N, d.OA, d.IA, d.B = N, OA, IA, B
    d.SA, d.da0, d.NTR, d.NTC, d.db0, d.db_step = form.SA, form.da0, form.NTR, form.NTC, form.db0, form.db_step
    d.M, d.Cq, d.CW, d.RW, d.WK = form.M, form.Cq, form.CW, form.RW, form.WK
    PW = form.PW if B % form.PW == 0 else 64
    d.PW = PW
    d.S = form.splits(N * (B // PW) * OA)
    d.ahead, d.bias = min(form.ahead, int(_os.environ.get("RFX_CLW_AHEAD", "99"))), int(form.bias)
    L = _lib.lib()
    nws = L.rfx_cl_wgrad_ws_floats(C.byref(d))
    if nws <= 0:
        raise RuntimeError("rfx_cl_wgrad: geometry rejected")
    ws = torch.empty(nws, device=p.device, dtype=torch.float32)
    d.ws = ws.data_ptr()
    check(L.rfx_cl_wgrad(C.byref(d), _stream()), "rfx_cl_wgrad")
    mp = _dev_map(form, p.device)
    check(L.rfx_cl_wgrad_reduce(C.c_void_p(ws.data_ptr()), C.c_void_p(mp.data_ptr()), mp.numel(), d.S, form.DT, form.RW, form.WK,
                                C.c_void_p(dw.data_ptr()), form.wn, C.c_void_p(db.data_ptr()) if db is not None else None,
                                int(accumulate), _stream()), "rfx_cl_wgrad_reduce")


def wform_conv(Cout, Cin, KA, KB):
    """dW of Conv2d(Cin -> Cout, (KA, KB), stride 1, padding same): P = output gradient (Cout), Q = input (Cin); weight (Cout, Cin, KA, KB)."""
    def widx(m, r, t, c):
        return ((m * Cin + c) * KA + r) * KB + t
    return WgradForm(Cout, Cin, KA, KB, 1, -(KA // 2), -(KB // 2), 1, widx, Cout * Cin * KA * KB)


def wform_conv_s4(Cout, Cin):
    """dW of Conv2d(Cin -> Cout, (8, 1), stride (4, 1), padding (2, 0)): P = output gradient, Q = input rows 4 oa + k - 2."""
    def widx(m, r, t, c):
        return (m * Cin + c) * 8 + r
    return WgradForm(Cout, Cin, 8, 1, 4, -2, 0, 0, widx, Cout * Cin * 8)


def wform_convtr_s4(Cin, Cout):
    """dW of ConvTranspose2d(Cin -> Cout, (8, 1), stride (4, 1)) cropped by 2 rows: P = layer input (Cin), Q = output gradient rows
    4 ia + k - 2 (Cout); weight (Cin, Cout, 8, 1).  The bias gradient (a sum of Q) is not this GEMM's: bias=False."""
    def widx(m, r, t, c):
        return (m * Cout + c) * 8 + r
    return WgradForm(Cin, Cout, 8, 1, 4, -2, 0, 0, widx, Cin * Cout * 8, bias=False)


def wform_conv_s4_fold(Cout, Cin):
    """dW of Conv1d(Cin -> Cout, 8, stride 4, padding 2): P = output gradient (.., L / 4, Cout), Q = input, folded (.., L / 4, 4 Cin)."""
    def widx(m, r, t, c):
        j, ci = c // Cin, c % Cin
        k = _k_down(t, j)
        return np.where((k >= 0) & (k < 8), (m * Cin + ci) * 8 + np.clip(k, 0, 7), -1)
    return WgradForm(Cout, 4 * Cin, 1, 3, 1, 0, -1, 1, widx, Cout * Cin * 8)


def wform_convtr_fold(Cin, Cout):
    """dW of the cropped ConvTranspose1d(Cin -> Cout, 8, stride 4): P = its input (.., L, Cin), Q = output gradient, folded (.., L, 4 Cout)."""
    def widx(m, r, t, c):
        j, co = c // Cout, c % Cout
        k = _k_down(t, j)
        return np.where((k >= 0) & (k < 8), (m * Cout + co) * 8 + np.clip(k, 0, 7), -1)
    return WgradForm(Cin, 4 * Cout, 1, 3, 1, 0, -1, 1, widx, Cin * Cout * 8, bias=False)
