"""Host-side planner: turns every convolution of the hot path into gather-GEMM
descriptors (include/remfx_hip.h: rfx_gemm_desc + rfx_ktab_entry tables).

All tensors are viewed as 4-D (N, C, A, B); 1-D convolutions use A == 1.
A plan is pure host data (numpy); ``remfx_amd.ops`` uploads the tables once per
(shape, stride) key and caches them.

Covered (reference call sites):
  conv forward            F.conv1d/conv2d        tcn.py:50,54,129; HDemucs enc; Cnn14; DCUNet
  conv input gradient     autograd of the above  (one descriptor per stride phase)
  conv_transpose forward  HDemucs / DCUNet decoders (one descriptor per stride phase, crop folded in)
  conv_transpose dgrad    = plain strided conv of the output gradient
  weight gradients        rfx_gemm_wgrad over the forward (conv) / dgrad (conv_transpose) descriptor
"""
from dataclasses import dataclass, field

import os

import numpy as np

INVALID_DA = -(1 << 30)


R_MAX = 4            # channel tiles per wave (capping it at 3 / 2 / 1 measured 168.8 / 175.4 / 198.8 vs 166.3 ms on the Demucs step;
                     # r03, with R = 3 at four waves per SIMD: 96-row tiles for every M % 96 == 0 still lose, 146.9 vs 144.3 ms)


def pick_r(M, K=1 << 30):
    """Mirror of rfx_gemm_pick_r (csrc/gemm.hip): channel tiles per wave."""
    if M <= 8:
        return 0
    if M <= 32:
        return 1
    if K <= 64:
        return 1
    if R_MAX < 4:                                     # experiment switch (RFX_R_MAX): cap the channel tiles per wave
        cands = [r for r in (3, 2) if r <= R_MAX] or [1]
        best = min(cands, key=lambda r: (-(-M // (32 * r)) * 32 * r, -r))
        return best
    best, best_pad = 4, -(-M // 128) * 128
    for r in (3, 2):
        pad = -(-M // (32 * r)) * 32 * r
        if pad < best_pad:
            best, best_pad = r, pad
    return best


def mpad_for(M, K=1 << 30):
    r = pick_r(M, K)
    return 8 if r == 0 else -(-M // (32 * r)) * 32 * r


@dataclass
class GemmPlan:
    N: int
    M: int
    K: int
    OA: int
    OB: int
    IA: int
    IB: int
    SA: int
    SB: int
    in_ns: int
    in_as: int
    in_bs: int
    out_ns: int
    out_cs: int
    out_as: int
    out_bs: int
    out_a0: int = 0
    out_b0: int = 0
    out_sa: int = 1
    out_sb: int = 1
    ktab: np.ndarray = None      # [Kpad, 4] int32 (off, da, db, flags)
    woff: np.ndarray = None      # [K] int32 weight gather offsets
    w_ms: int = 0                # weight stride per output row m
    Mpad: int = 0
    Kpad: int = 0
    R: int = 0
    mg_log: int = 0
    mg_axis: int = 0
    mg_len: int = 0
    mg_off: int = 0
    # tap-major form (built by finalize when cin >= 8; consumed by the bf16x3 / bf16 MFMA kernels):
    # the reduction axis is re-ordered (tap, channel) in groups of 8 channels, see finalize()
    cin: int = 0                 # channels of the gathered operand (rows of ktab are channel-major: k = ci*ntaps + t)
    in_cs: int = 0               # its channel stride (elements)
    ntaps: int = 0
    gpt: int = 0                 # 8-channel groups per tap = ceil(cin / 8)
    Kpad_t: int = 0              # 16 * ceil(ntaps * gpt / 2)
    in_extent: int = 0           # elements spanned by one sample of the operand (buffer range check)
    tap_tab: np.ndarray = None   # [ntaps + 16, 4] int32 (off of channel 0, da, db, 0); tail rows invalid
    woff_t: np.ndarray = None    # [Kpad_t] int32 weight gather offsets in tap-major order, -1 = zero row
    # halo-tile form (csrc/gemm_halo.h; set by _halo_geometry when the plan qualifies, all 0 otherwise)
    halo_nt: int = 0             # real taps (3 or 9)
    halo_rows: int = 0           # input rows the taps of one output row touch
    halo_w: int = 0              # input columns the taps of a 256-position tile touch
    halo_da0: int = 0            # first row / column displacement
    halo_db0: int = 0
    halo_pad: int = 0
    extra: dict = field(default_factory=dict)

    def finalize(self, bias_row=False):
        """Pad the tables.  bias_row appends a constant-one column (flags bit0) used
        by the weight-gradient kernel to produce the bias gradient."""
        kt = np.asarray(self.ktab, dtype=np.int64).reshape(-1, 4)
        K = kt.shape[0]
        self._build_tap_major(kt, np.asarray(self.woff, dtype=np.int64).reshape(-1))
        if bias_row:
            kt = np.concatenate([kt, np.array([[0, 0, 0, 1]], dtype=np.int64)], 0)
        Kall = kt.shape[0]
        self.K = Kall
        self.Kpad = -(-Kall // 16) * 16
        # + 6 extra all-invalid K steps: the MFMA kernel prefetches table rows of step
        # ks+4 and gathers of step ks+3 unconditionally (branch-free main loop)
        pad = np.zeros((self.Kpad + 96 - Kall, 4), dtype=np.int64)
        pad[:, 1] = INVALID_DA
        kt = np.concatenate([kt, pad], 0)
        assert np.abs(kt[:, 0]).max(initial=0) < 2 ** 31
        # the bf16x3 kernels address one sample through a 32-bit byte offset (raw buffer loads, 2 GiB window)
        reach = (int(kt[:K, 0].max(initial=0)) + (self.IA - 1) * abs(int(self.in_as))
                 + (self.IB - 1) * abs(int(self.in_bs)))
        oreach = ((self.M - 1) * abs(int(self.out_cs)) + (self.OA - 1) * max(self.out_sa, 1) * abs(int(self.out_as))
                  + (self.OB - 1) * max(self.out_sb, 1) * abs(int(self.out_bs)))
        if reach * 4 >= 2 ** 31 or oreach * 4 >= 2 ** 31:
            raise ValueError("gather-GEMM: one sample of the input operand must span < 2 GiB")
        self.ktab = kt.astype(np.int32)
        self.R = pick_r(self.M, K)
        self.Mpad = mpad_for(self.M, K)
        self.woff = np.asarray(self.woff, dtype=np.int32).reshape(-1)
        assert self.woff.shape[0] == K
        self.extra["n_weight_rows"] = K
        return self


TAP_BLOCK_GROUPS = 2      # smallest channel block in 8-channel groups; 0 = no blocking


def _tap_major(self, kt, woff):
    """Tap-major re-ordering of the reduction axis for the bf16 MFMA kernels (csrc/gemm_tap.h).

    The builders emit rows channel-major, k = ci * ntaps + t with off = ci * in_cs + tapoff[t]: every row of a tap
    shares (da, db).  The kernels walk 8-CHANNEL GROUPS g = t * gpt + c8 (gpt = ceil(cin / 8)); MFMA K step ks takes
    group 2ks on lanes 0-31 and group 2ks + 1 on lanes 32-63, so a lane does ONE bounds test and one offset add per
    8 gathers (the channel-major table cost two tests and a select per gather) and the 8 loads differ only by a
    multiple of the channel stride.  Channels >= cin of the last group read past the sample's extent (hardware
    buffer range check -> 0) and meet zero weight rows (woff_t = -1)."""
    K = kt.shape[0]
    if self.cin > 0:                   # exact span of one sample of the operand: the num_records of its buffer descriptor
        self.in_extent = ((self.cin - 1) * abs(int(self.in_cs)) + (self.IA - 1) * abs(int(self.in_as))
                          + (self.IB - 1) * abs(int(self.in_bs)) + 1)
    if self.cin < 8 or K == 0 or K % self.cin:
        self.cin = 0
        return
    nt = K // self.cin
    if nt > 112:                       # the kernels keep the whole tap table in LDS (128 slots incl. padding)
        self.cin = 0
        return
    taps = kt[:nt].copy()
    ci = np.arange(self.cin)[:, None]
    exp_off = (taps[None, :, 0] + ci * self.in_cs).reshape(-1)
    if not (np.array_equal(kt[:, 0], exp_off) and np.array_equal(kt[:, 1:3], np.tile(taps[:, 1:3], (self.cin, 1)))
            and not kt[:, 3].any()):
        raise AssertionError("gather-GEMM plan rows are not (channel, tap) ordered")
    gpt_all = -(-self.cin // 8)
    # Channel blocking: with many channels AND several taps, "all channels of tap 0, then all channels of tap 1, ..." re-reads a
    # position tile's input once per tap with a whole channel sweep (hundreds of KB per workgroup) in between: the r02 per-launch
    # PMC join measured 5-7.5x the algorithmic bytes on the 3x3 layers (192 ch x 9 taps: 11.2 GB read for a 1.6 GB operand).
    # The reduction is therefore cut into blocks of gb groups (8 gb channels); inside a block the taps are walked back to back,
    # so the re-use distance is one block (16 channels x rows x 130 positions ~ 25 KB).  The kernels do not know: a block is
    # presented as nt "virtual taps" whose table offset includes the block's channel offset, and gpt = gb.
    gb, nb = gpt_all, 1
    if nt > 1 and gpt_all >= 4 and TAP_BLOCK_GROUPS > 0:
        fits = [c for c in range(TAP_BLOCK_GROUPS, gpt_all) if nt * -(-gpt_all // c) <= 112]
        even = [c for c in fits if gpt_all % c == 0]
        pick = (even or fits or [gpt_all])[0]
        if even and fits and even[0] > 2 * fits[0]:         # a divisor far above the smallest fitting block: accept the padding
            pick = fits[0]
        gb, nb = pick, -(-gpt_all // pick)
    ntv = nt * nb
    G = ntv * gb
    self.ntaps, self.gpt, self.Kpad_t = ntv, gb, 16 * (-(-G // 2))
    tab = np.zeros((ntv + 16, 4), dtype=np.int64)
    for cb in range(nb):
        tab[cb * nt:(cb + 1) * nt] = taps
        tab[cb * nt:(cb + 1) * nt, 0] += cb * gb * 8 * int(self.in_cs)
    tab[ntv:, 1] = INVALID_DA
    if np.abs(tab[:, 0]).max() > 0x3fffffff:
        raise AssertionError("tap offset does not fit 32 bits")
    self.tap_tab = tab.astype(np.int32)
    wt = np.full(self.Kpad_t, -1, dtype=np.int64)
    cbi, t, c = np.meshgrid(np.arange(nb), np.arange(nt), np.arange(gb * 8), indexing="ij")
    ch = cbi * gb * 8 + c
    ok = ch < self.cin
    wt[(((cbi * nt + t) * gb * 8) + c)[ok]] = woff[(ch * nt + t)[ok]]
    self.woff_t = wt.astype(np.int32)
    self._halo_geometry(taps, nt, gb)


HALO = True               # bench.py --no-halo / tests switch the halo-tile kernel off (plans built while it is False carry halo_nt = 0)
HALO_MIN_POSITIONS = 1 << 19
HALO_TW = 128             # output positions per workgroup of gemm_halo_kernel (csrc/gemm_halo.h: RFX_HALO_TW)


def _halo_geometry(self, taps, nt, gb):
    """Does gemm_halo_kernel take this plan?  Stride-1 convolution with 3 or 9 real taps over a unit-stride position axis whose
    length is a multiple of the 128-position tile, channel blocks of an even number of 8-channel groups (a K step never straddles two
    taps), and a halo tile that the kernel's staging slots cover: rows <= 3, rows x columns <= 256 (3 taps) / 512 (9 taps).
    The first nt table rows are the real taps; the kernel derives each tap's position shift inside the tile from (da, db)."""
    self.halo_nt = self.halo_rows = self.halo_w = self.halo_da0 = self.halo_db0 = 0
    if not HALO or nt not in (3, 9) or gb < 2 or gb % 2 or self.SA != 1 or self.SB != 1 or int(self.in_bs) != 1:
        return
    if self.OB % HALO_TW or self.Kpad_t != 8 * self.ntaps * self.gpt:
        return
    # few positions, many output channels (the deep layers: 64 x 128 ... 64 x 4096 positions): every 128-position workgroup re-reads
    # its whole A chunk and the tap-major kernel is as fast or faster (scripts/perf_halo.py: 0.43 vs 0.47 ms at 1536 -> 3072 x 128)
    if self.N * self.OA * self.OB < HALO_MIN_POSITIONS:
        return
    da, db = taps[:, 1], taps[:, 2]
    if not np.array_equal(taps[:, 0], da * int(self.in_as) + db * int(self.in_bs)):
        return
    rows, w = int(da.max() - da.min()) + 1, HALO_TW + int(db.max() - db.min())
    if rows > 3 or rows * w > 256 * (2 if nt == 9 else 1):
        return
    self.halo_nt, self.halo_rows, self.halo_w, self.halo_da0, self.halo_db0 = nt, rows, w, int(da.min()), int(db.min())


GemmPlan._halo_geometry = _halo_geometry


GemmPlan._build_tap_major = _tap_major


def _out_len(i, k, s, p, d):
    return (i + 2 * p - d * (k - 1) - 1) // s + 1


PRUNE_TAPS = os.environ.get("RFX_PRUNE_TAPS", "1") != "0"      # A/B: 0 keeps the padding-only taps in the reduction


def _live_taps(I, O, K, S, P, D):
    """Taps k of one axis that meet the operand for at least one output index: o * S - P + k * D in [0, I) for some o in [0, O).
    (All of them unless the operand is shorter than the kernel's reach; never empty for a valid convolution.)"""
    if not PRUNE_TAPS:
        return np.arange(K)
    idx = np.arange(O)[:, None] * S - P + np.arange(K)[None, :] * D
    live = np.nonzero(((idx >= 0) & (idx < I)).any(0))[0]
    return live if len(live) else np.arange(K)


def conv_fwd_plan(xshape, xstrides, wshape, stride, padding, dilation, ystrides, bias_row=False):
    """Conv2d forward:  y[n,co,oa,ob] = sum w[co,ci,ka,kb] x[n,ci,oa*SA-PA+ka*DA, ob*SB-PB+kb*DB]."""
    N, Cin, IA, IB = xshape
    Cout, Cin_w, KA, KB = wshape
    assert Cin_w == Cin
    (SA, SB), (PA, PB), (DA, DB) = stride, padding, dilation
    OA, OB = _out_len(IA, KA, SA, PA, DA), _out_len(IB, KB, SB, PB, DB)
    ns, cs, as_, bs = xstrides
    # taps that only ever meet the zero padding are left out of the reduction (a 3 x 3 rewrite over the ONE frequency row the
    # deepest frequency layer of Hybrid Demucs keeps has 6 such taps of 9: a third of the products remain); their weight
    # gradient is exactly zero, so the weight-gradient GEMM built on the same plan skips them too
    la, lb = _live_taps(IA, OA, KA, SA, PA, DA), _live_taps(IB, OB, KB, SB, PB, DB)
    ci, ka, kb = np.meshgrid(np.arange(Cin), la, lb, indexing="ij")
    da, db = ka * DA - PA, kb * DB - PB
    ktab = np.stack([ci * cs + da * as_ + db * bs, da, db, np.zeros_like(da)], -1).reshape(-1, 4)
    woff = (ci * KA * KB + ka * KB + kb).reshape(-1)
    dense = len(la) == KA and len(lb) == KB
    p = GemmPlan(N=N, M=Cout, K=ktab.shape[0], OA=OA, OB=OB, IA=IA, IB=IB, SA=SA, SB=SB,
                 in_ns=ns, in_as=as_, in_bs=bs, out_ns=ystrides[0], out_cs=ystrides[1],
                 out_as=ystrides[2], out_bs=ystrides[3], ktab=ktab, woff=woff, w_ms=Cin * KA * KB, cin=Cin, in_cs=cs)
    p.extra["out_shape"] = (N, Cout, OA, OB)
    p.extra["dense"] = dense             # False: some (ka, kb) have no row (their weight gradient is zero)
    return p.finalize(bias_row)


def _phase_taps(phi, K, S, P, D):
    """taps k with (phi + P - k*D) % S == 0 and their input shift c = (phi + P - k*D) // S."""
    return [(k, (phi + P - k * D) // S) for k in range(K) if (phi + P - k * D) % S == 0]


def conv_dgrad_plans(xshape, xstrides, wshape, stride, padding, dilation, gshape, gstrides):
    """Input gradient of conv_fwd_plan: one descriptor per stride phase.
    dx[n,ci,ia,ib] = sum_{co,ka,kb} w[co,ci,ka,kb] g[n,co,(ia+PA-ka*DA)/SA,(ib+PB-kb*DB)/SB]."""
    N, Cin, IA, IB = xshape
    Cout, _, KA, KB = wshape
    (SA, SB), (PA, PB), (DA, DB) = stride, padding, dilation
    _, _, OA, OB = gshape
    gns, gcs, gas, gbs = gstrides
    plans = []
    for pa in range(min(SA, IA)):
        QA = -(-(IA - pa) // SA)
        ta = [(k, c) for (k, c) in _phase_taps(pa, KA, SA, PA, DA) if -QA < c < OA or not PRUNE_TAPS]      # q + c in [0, OA) for some q in [0, QA)
        for pb in range(min(SB, IB)):
            QB = -(-(IB - pb) // SB)
            tb = [(k, c) for (k, c) in _phase_taps(pb, KB, SB, PB, DB) if -QB < c < OB or not PRUNE_TAPS]
            rows, woff = [], []
            for co in range(Cout):
                for (ka, ca) in ta:
                    for (kb, cb) in tb:
                        rows.append((co * gcs + ca * gas + cb * gbs, ca, cb, 0))
                        woff.append(co * Cin * KA * KB + ka * KB + kb)
            p = GemmPlan(N=N, M=Cin, K=len(rows), OA=QA, OB=QB, IA=OA, IB=OB, SA=1, SB=1,
                         in_ns=gns, in_as=gas, in_bs=gbs, out_ns=xstrides[0], out_cs=xstrides[1],
                         out_as=xstrides[2], out_bs=xstrides[3], out_a0=pa, out_b0=pb, out_sa=SA,
                         out_sb=SB, ktab=np.array(rows, dtype=np.int64).reshape(-1, 4), woff=woff,
                         w_ms=KA * KB, cin=Cout if ta and tb else 0, in_cs=gcs)
            plans.append(p.finalize())
    return plans


def convT_out_len(i, k, s, d, output_padding=0):
    return (i - 1) * s + d * (k - 1) + 1 + output_padding


def convT_fwd_plans(xshape, xstrides, wshape, stride, dilation, crop_lo, out_len, ystrides):
    """ConvTranspose2d forward with the output crop folded in (HDemucs decoders:
    z[..., pad:pad+length]; padding=p of nn.ConvTranspose is the same thing with
    crop_lo = p).  w: [Cin][Cout][KA][KB].
    y[n,co,o_a,o_b] = sum_{ci,ka,kb} w[ci,co,ka,kb] x[n,ci,(o_a+lo_a-ka*DA)/SA, ...]."""
    N, Cin, IA, IB = xshape
    Cin_w, Cout, KA, KB = wshape
    assert Cin_w == Cin
    (SA, SB), (DA, DB) = stride, dilation
    (la, lb), (LA, LB) = crop_lo, out_len
    ns, cs, as_, bs = xstrides

    def phase(phi, K, S, D, lo, L):
        qmin = max(0, -(-(lo - phi) // S))
        qmax = (L - 1 + lo - phi) // S
        taps = [(k, qmin + (phi - k * D) // S) for k in range(K) if (phi - k * D) % S == 0]
        return qmax - qmin + 1, S * qmin + phi - lo, taps

    plans = []
    for pa in range(SA):
        QA, a0, ta = phase(pa, KA, SA, DA, la, LA)
        if QA <= 0:
            continue
        for pb in range(SB):
            QB, b0, tb = phase(pb, KB, SB, DB, lb, LB)
            if QB <= 0:
                continue
            rows, woff = [], []
            for ci in range(Cin):
                for (ka, da) in ta:
                    for (kb, db) in tb:
                        rows.append((ci * cs + da * as_ + db * bs, da, db, 0))
                        woff.append(ci * Cout * KA * KB + ka * KB + kb)
            p = GemmPlan(N=N, M=Cout, K=len(rows), OA=QA, OB=QB, IA=IA, IB=IB, SA=1, SB=1,
                         in_ns=ns, in_as=as_, in_bs=bs, out_ns=ystrides[0], out_cs=ystrides[1],
                         out_as=ystrides[2], out_bs=ystrides[3], out_a0=a0, out_b0=b0, out_sa=SA,
                         out_sb=SB, ktab=np.array(rows, dtype=np.int64).reshape(-1, 4), woff=woff,
                         w_ms=KA * KB, cin=Cin if ta and tb else 0, in_cs=cs)
            plans.append(p.finalize())
    return plans


def convT_dgrad_plan(xshape, xstrides, wshape, stride, dilation, crop_lo, gshape, gstrides,
                     bias_row=False):
    """Input gradient of convT_fwd_plans = a strided conv of g with padding crop_lo:
    dx[n,ci,ia,ib] = sum_{co,ka,kb} w[ci,co,ka,kb] g[n,co,ia*SA+ka*DA-lo_a, ib*SB+kb*DB-lo_b].
    Its weight-gradient (rfx_gemm_wgrad with in=g, gout=x) yields dW of the transposed conv."""
    N, Cin, IA, IB = xshape
    _, Cout, KA, KB = wshape
    (SA, SB), (DA, DB) = stride, dilation
    la, lb = crop_lo
    _, _, GA, GB = gshape
    gns, gcs, gas, gbs = gstrides
    co, ka, kb = np.meshgrid(np.arange(Cout), np.arange(KA), np.arange(KB), indexing="ij")
    da, db = ka * DA - la, kb * DB - lb
    ktab = np.stack([co * gcs + da * gas + db * gbs, da, db, np.zeros_like(da)], -1).reshape(-1, 4)
    woff = (co * KA * KB + ka * KB + kb).reshape(-1)
    p = GemmPlan(N=N, M=Cin, K=ktab.shape[0], OA=IA, OB=IB, IA=GA, IB=GB, SA=SA, SB=SB,
                 in_ns=gns, in_as=gas, in_bs=gbs, out_ns=xstrides[0], out_cs=xstrides[1],
                 out_as=xstrides[2], out_bs=xstrides[3], ktab=ktab, woff=woff, w_ms=Cout * KA * KB, cin=Cout, in_cs=gcs)
    return p.finalize(bias_row)


def shift_plan(xshape, xstrides, M, shift, oshape, ostrides):
    """1x1 conv reading x[..., b + shift] (out-of-range -> 0): ktab rows (ci*cs + shift*bs, 0, shift).
    Used for the TCN residual crop (tcn.py:54-58) and the LSTM dW_hh product (h_{t-1} = out shifted by Bn)."""
    N, Cin, IA, IB = xshape
    ns, cs, as_, bs = xstrides
    ci = np.arange(Cin)
    ktab = np.stack([ci * cs + shift * bs, np.zeros_like(ci), np.full_like(ci, shift), np.zeros_like(ci)], -1)
    on, oc, oa, ob = ostrides
    p = GemmPlan(N=N, M=M, K=Cin, OA=oshape[2], OB=oshape[3], IA=IA, IB=IB, SA=1, SB=1,
                 in_ns=ns, in_as=as_, in_bs=bs, out_ns=on, out_cs=oc, out_as=oa, out_bs=ob,
                 ktab=ktab, woff=ci.copy(), w_ms=Cin, cin=Cin, in_cs=cs)
    return p.finalize()


def merged_phase_plan(inshape, instrides, rows, axis, G, J, off, out_len, ostrides):
    """All G stride phases of a transposed convolution / strided-conv input gradient as ONE GEMM.

    The operand `in` (N, Cin, IA, IB) is correlated with J = K/G taps along `axis` (full correlation: positions
    i = 0 .. I+J-2, tap t reads in[i + t - (J-1)]); `rows` = G * Cout output rows ordered (channel, phase).  Row
    (c, q) of position i lands at output axis index i*G + q + off (rfx_gemm_desc.mg_*).  The caller supplies the
    weight as a dense (rows, Cin, J) tensor.  Positions are extended so that every output index below out_len is
    written (taps outside the operand read as zero)."""
    N, Cin, IA, IB = inshape
    I = (IA, IB)[axis]
    npos = max(I + J - 1, -(-(out_len - off) // G))
    K = (J, 1) if axis == 0 else (1, J)
    pad = (J - 1, 0) if axis == 0 else (0, J - 1)
    p = conv_fwd_plan(inshape, instrides, (rows, Cin) + K, (1, 1), pad, (1, 1), ostrides)
    if axis == 0:
        p.OA = npos
    else:
        p.OB = npos
    p.mg_log, p.mg_axis, p.mg_len, p.mg_off = int(G).bit_length() - 1, axis, out_len, off
    p.halo_nt = 0                # the merged store is not a halo-kernel epilogue (and OA / OB were just changed)
    if p.R == 0:                 # few output rows (last decoders: 1-2 channels x 4 phases): the thin kernel has no merged store;
        p.R, p.Mpad = 1, 32      # one MFMA launch that reads the operand once beats G thin launches that each re-read it
    p.extra["out_shape"] = None
    return p
