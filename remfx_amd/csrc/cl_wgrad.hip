// Weight gradients on channels-last bf16 operands (rfx_cl_wgrad + rfx_cl_wgrad_reduce): the frequency branch of Hybrid Demucs in
// the bf16 arithmetic mode -- dW of torchaudio HDemucs `_HEncLayer.conv / rewrite` and `_HDecLayer.rewrite / conv_tr`, reached from
// remfx/models.py:308,317 through loss.backward().
//
//   D[m][(r, t, c)] = sum over (n, oa, b) of  P[n][oa][b][m] * Q[n][oa * SA + da0 + r][b + db0 + t * db_step][c]
//
// The reduction runs over POSITIONS, the slow axis of both operands, so neither MFMA operand is contiguous in memory.  Both go
// global -> LDS by DMA as dense [position][channels] images (contiguous source, contiguous destination, no VGPRs) and are read
// back with `ds_read_b64_tr_b16`: a 16-lane group hands in the addresses of 4 positions x 16 channels (8 bytes per lane) and every
// lane receives ITS channel at the 4 positions -- two such reads are one 32x32x16 MFMA fragment, A and B alike.
//
// Work decomposition.  The K space is cut into STEPS of 64 consecutive positions of one (n, oa) row; a workgroup (8 waves) owns a
// contiguous range of steps (one of S position splits) and one D tile = 32 RW rows x up to 16 column tiles of 32 = (tap, 16-channel
// group) pairs of a CW-channel slice of Q.  Waves are arranged WC x WK: WC column groups of 2 tiles each, WK-way split of the four
// K steps of a piece (small D tiles put their waves on K instead of on columns).  Per step ONE P piece (64 x 32 RW channels) and the
// SA NEW rows of Q (64 + halo positions x CW channels) arrive; Q rows live in a ring, so a row is fetched once per workgroup however
// many row taps read it.  A range starts with PRE load-only steps that fill the ring.  Pieces are spread evenly over the waves
// (PPW per wave and step, surplus slots re-fetch the last piece) so one counted `s_waitcnt vmcnt` per step covers a wave's share;
// the loads run AHEAD steps in front of the MFMAs.
//
// Determinism: every workgroup dumps its accumulators into its own workspace slot; rfx_cl_wgrad_reduce sums the slots in a fixed
// order and scatters through a host-built index map into the weight (and bias: a column tile fed with ones) gradient.  No atomics.
#include "cl_common.h"

// ablation builds (scripts/build_abl.py cl_wgrad RFX_CLW_DBG_BUILD n, dev only): 1 no DMA, 2 no fragment reads, 4 no MFMA, 8 no barrier,
// 16 no staggering of the two waves of a SIMD.  Compile-time: as run-time switches they put a select in front of every fragment read.
#ifndef RFX_CLW_DBG_BUILD
#define RFX_CLW_DBG_BUILD 0
#endif
#define CLW_DBG RFX_CLW_DBG_BUILD
#include <stdlib.h>

struct ClWgK {
  rfx_cl_wgrad_desc d;
  int32_t nbq, MTn, CTn, DT, spx;
  int32_t total, sps;                 // real steps in all, per split
  int32_t NP, NQ, TP, PPW;            // P pieces, Q pieces per row, pieces per step, pieces per wave and step
  int32_t PSLOT, QROWB, QROWP, R, PD, PRE, HB;
  int32_t HN, NCG, bias_tile;
  uint32_t p_bytes, q_bytes;
};

#define CL_WG_MAXP 6

__device__ __forceinline__ void clw_wait_vm(int n) {
  switch (n) {
    case 0: CL_VMCNT(0); break;   case 1: CL_VMCNT(1); break;   case 2: CL_VMCNT(2); break;   case 3: CL_VMCNT(3); break;
    case 4: CL_VMCNT(4); break;   case 5: CL_VMCNT(5); break;   case 6: CL_VMCNT(6); break;   case 7: CL_VMCNT(7); break;
    case 8: CL_VMCNT(8); break;   case 9: CL_VMCNT(9); break;   case 10: CL_VMCNT(10); break; case 11: CL_VMCNT(11); break;
    case 12: CL_VMCNT(12); break; case 13: CL_VMCNT(13); break; case 14: CL_VMCNT(14); break; case 15: CL_VMCNT(15); break;
    case 16: CL_VMCNT(16); break; case 17: CL_VMCNT(17); break; case 18: CL_VMCNT(18); break; case 19: CL_VMCNT(19); break;
    case 20: CL_VMCNT(20); break; case 21: CL_VMCNT(21); break; case 22: CL_VMCNT(22); break; case 23: CL_VMCNT(23); break;
    default: CL_VMCNT(24); break;
  }
}

typedef short clw_s16x4 __attribute__((ext_vector_type(4)));
typedef short clw_s16x8 __attribute__((ext_vector_type(8)));
// one MFMA fragment (8 k values of this lane's row / column) = two transposing reads 4 positions apart
__device__ __forceinline__ cl_bf16x8 clw_frag(const unsigned char* p, int step4) {
  const clw_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) clw_s16x4*)(p));
  const clw_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) clw_s16x4*)(p + step4));
  const clw_s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(cl_bf16x8, v);
}

struct ClwIt {                       // walks the steps of a range: PRE load-only steps in front of the range and of every column
  int n, bq, oa, pre;                // column = (sample n, position block bq)
};

template <int RW, int WK, int PW>
__global__ __launch_bounds__(512, 2) void cl_wgrad_kernel(const ClWgK g) {
  constexpr int NT = 2, WC = 8 / WK, KSW = (PW / 16) / WK;      // PW positions per step
  constexpr int PROWB = 64 * RW;                               // bytes of one position in the P image
  extern __shared__ __attribute__((aligned(16))) unsigned char clw_smem[];
  const rfx_cl_wgrad_desc& d = g.d;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int kq = wave / WC, cw = wave % WC;
  // the D tiles of one position split read the same P / Q bytes: they sit on ONE XCD (one L2), adjacent in dispatch order
  // (fewer than 8 splits: plain order, the D tiles of a split spread over the XCDs -- grouping them would leave XCDs idle)
  const int bid = blockIdx.x, xcd = bid & 7, qx = bid >> 3;
  const bool grouped = d.S >= 8;
  const int dt = grouped ? qx % g.DT : bid % g.DT, split = grouped ? xcd * g.spx + qx / g.DT : bid / g.DT;
  if (split >= d.S) return;
  const int mt = dt / g.CTn, ct = dt - mt * g.CTn;
  const int tau0 = split * g.sps;
  const int tau1 = min(tau0 + g.sps, g.total);
  const int OA = d.OA;

  unsigned char* const pbase_lds = clw_smem;
  unsigned char* const qbase_lds = clw_smem + g.PD * g.PSLOT;

  // ---- this wave's share of the pieces of a step
  int32_t rel[CL_WG_MAXP], bpos[CL_WG_MAXP];
  int lds_off[CL_WG_MAXP], prow[CL_WG_MAXP];                   // prow: -1 = P piece, else the new-row index rr of a Q piece
#pragma unroll
  for (int i = 0; i < CL_WG_MAXP; ++i) {
    int e = wave + 8 * i;
    e = e < g.TP ? e : g.TP - 1;
    if (e < g.NP) {
      prow[i] = -1;
      lds_off[i] = e * 1024;
      const int o = e * 1024 + lane * 16;
      const int pos = o / PROWB, cb = o - pos * PROWB;
      const bool ok = pos < PW && (mt * 32 * RW + (cb >> 1)) < d.M;
      rel[i] = ok ? (int32_t)((pos * d.p.bs + d.p.c0 + mt * 32 * RW) * 2 + cb) : (int32_t)CL_OOB;
      bpos[i] = 0;
    } else {
      const int qe = e - g.NP;
      const int rr = qe / g.NQ, qi = qe - rr * g.NQ;
      prow[i] = rr;
      lds_off[i] = rr * g.QROWB + qi * 1024;
      const int o = qi * 1024 + lane * 16;
      const int pos = o / g.QROWP, cb = o - pos * g.QROWP;
      const bool ok = pos < PW + 2 * g.HB && (ct * d.CW + (cb >> 1)) < d.Cq;
      rel[i] = ok ? (int32_t)(((pos - g.HB) * d.q.bs + d.q.c0 + ct * d.CW) * 2 + cb) : (int32_t)CL_OOB;
      bpos[i] = ok ? pos - g.HB : 0x40000000;
    }
  }

  // ---- fragment addresses of this lane
  const int gq = lane >> 4, i16 = lane & 15;
  const int rowhalf = gq & 1, khalf = gq >> 1, s_e = i16 >> 2, q4 = i16 & 3;
  const uint32_t a0 = (uint32_t)((8 * khalf + s_e) * PROWB + rowhalf * 32 + q4 * 8);
  int rtap[NT];
  uint32_t bconst[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    int hh = 2 * (cw * NT + t) + rowhalf;
    hh = hh < g.HN ? hh : 0;                                   // unused halves compute on valid data, never stored
    const int cgl = hh % g.NCG, tt = hh / g.NCG;
    const int tc = tt % d.NTC;
    rtap[t] = tt / d.NTC;
    bconst[t] = (uint32_t)((g.HB + d.db0 + tc * d.db_step + 8 * khalf + s_e) * g.QROWP + cgl * 32 + q4 * 8);
  }
  const bool ones_wave = d.bias && g.bias_tile >= 0 && (g.bias_tile / NT) == cw;
  const int ones_t = g.bias_tile % NT;

  f32x16 acc[RW][NT];
#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;

  // ---- step sequence
  const int col0 = tau0 / OA;
  const int ncols = (tau1 - 1) / OA - col0 + 1;
  const int L = (tau1 - tau0) + g.PRE * ncols;
  ClwIt it_i = {col0 / g.nbq, col0 % g.nbq, tau0 - col0 * OA - g.PRE, g.PRE}, it_c = it_i;
  auto advance = [&](ClwIt& it) {
    ++it.oa;
    if (it.pre > 0) --it.pre;
    if (it.oa == OA) {
      it.oa = -g.PRE; it.pre = g.PRE;
      if (++it.bq == g.nbq) { it.bq = 0; ++it.n; }
    }
  };
  // sample descriptors: rebuilt only when the issue side moves to another sample
  // (the pieces are issued by asm the compiler does not track -- cl_glds16_quiet: with the builtin it drained the whole ring with a
  // vmcnt(0) in front of the first fragment read of every step, whatever the counted wait above the barrier had left in flight)
  int rs_n = it_i.n;
  cl_i32x4 rs_p = cl_rsrc_words(reinterpret_cast<const uint16_t*>(d.p.p) + (int64_t)rs_n * d.p.ns, g.p_bytes);
  cl_i32x4 rs_q = cl_rsrc_words(reinterpret_cast<const uint16_t*>(d.q.p) + (int64_t)rs_n * d.q.ns, g.q_bytes);
  int iu = 0, wp = 0, ps_i = 0;                                 // issue side: next step, ring write row, P slot
  auto issue_next = [&]() {
    if (CLW_DBG & 1) { advance(it_i); ++iu; return; }
    const int b0 = it_i.bq * PW;
    const bool real = it_i.pre == 0;
    if (it_i.n != rs_n) {
      rs_n = it_i.n;
      rs_p = cl_rsrc_words(reinterpret_cast<const uint16_t*>(d.p.p) + (int64_t)rs_n * d.p.ns, g.p_bytes);
      rs_q = cl_rsrc_words(reinterpret_cast<const uint16_t*>(d.q.p) + (int64_t)rs_n * d.q.ns, g.q_bytes);
    }
    const int32_t pb = (int32_t)(((int64_t)it_i.oa * d.p.as + (int64_t)b0 * d.p.bs) * 2);
    const int ia0 = it_i.oa * d.SA + d.da0 + d.NTR - d.SA;
#pragma unroll
    for (int i = 0; i < CL_WG_MAXP; ++i) {
      if (i < g.PPW) {
        if (prow[i] < 0) {
          const uint32_t vo = real ? (uint32_t)(rel[i] + pb) : CL_OOB;
          cl_glds16_quiet(rs_p, pbase_lds + ps_i * g.PSLOT + lds_off[i], vo);
        } else {
          const int ia = ia0 + prow[i];
          const bool rok = (unsigned)ia < (unsigned)d.IA;
          const int32_t qb = (int32_t)(((int64_t)ia * d.q.as + (int64_t)b0 * d.q.bs) * 2);
          const uint32_t vo = (rok && (unsigned)(b0 + bpos[i]) < (unsigned)d.B) ? (uint32_t)(rel[i] + qb) : CL_OOB;
          cl_glds16_quiet(rs_q, qbase_lds + wp * g.QROWB + lds_off[i], vo);
        }
      }
    }
    wp += d.SA;
    wp = wp >= g.R ? wp - g.R : wp;
    ps_i = ps_i + 1 == g.PD ? 0 : ps_i + 1;
    advance(it_i);
    ++iu;
  };

  int cpm = (d.SA - d.NTR + 2 * g.R) % g.R;                     // ring row of tap 0 at the current compute step
  int ps_c = 0;
  const int ahead = g.PD - 1;
  for (int i = 0; i < ahead; ++i)
    if (iu < L) issue_next();
  // Waves w and w + 4 share a SIMD.  After the barrier waves 0..3 issue their DMA pieces FIRST and then compute, waves 4..7 compute
  // first and issue at the END of the step: one wave of a SIMD feeds the matrix pipe while the other is busy issuing loads.  (With
  // all eight waves in the same phase nothing overlapped -- ablation builds, r05: 1.29 ms = skeleton 0.18 + DMA issue 0.29 + fragment
  // reads 0.25 + MFMA 0.58 for the 48 -> 96 3x3 layer, exactly additive.)  Both orders have issued groups 0 .. i + ahead - 1 when
  // they wait in step i, so the counted wait is the same.
  const bool late = wave >= 4 && !(CLW_DBG & 16);
  auto frags = [&](int ks, const unsigned char* pb, const unsigned char* const (&qb)[NT], cl_bf16x8 (&af)[RW], cl_bf16x8 (&bf)[NT], int i) {
#pragma unroll
    for (int r = 0; r < RW; ++r)
      af[r] = (CLW_DBG & 2) ? __builtin_bit_cast(cl_bf16x8, make_uint4(lane, r, ks, i)) : clw_frag(pb + ks * 16 * PROWB + r * 64, 4 * PROWB);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (ones_wave && t == ones_t) {
        const clw_s16x8 one = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
        bf[t] = __builtin_bit_cast(cl_bf16x8, one);
      } else {
        bf[t] = (CLW_DBG & 2) ? __builtin_bit_cast(cl_bf16x8, make_uint4(lane, t, ks, i)) : clw_frag(qb[t] + ks * 16 * g.QROWP, 4 * g.QROWP);
      }
    }
  };
  for (int i = 0; i < L; ++i) {
    const int left = L - 1 - i;
    clw_wait_vm((left < ahead - 1 ? left : ahead - 1) * g.PPW);
    if (!(CLW_DBG & 8)) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (!late && iu < L) issue_next();
    __builtin_amdgcn_sched_barrier(0);
    if (it_c.pre == 0 && !(CLW_DBG & 4)) {
      const unsigned char* pb = pbase_lds + ps_c * g.PSLOT + a0;
      const unsigned char* qb[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        int sl = cpm + rtap[t];
        sl = sl >= g.R ? sl - g.R : sl;
        qb[t] = qbase_lds + sl * g.QROWB + bconst[t];
      }
      // fragments of K step k + 1 are read while the MFMAs of step k run
      cl_bf16x8 af[2][RW], bf[2][NT];
      frags(kq, pb, qb, af[0], bf[0], i);
#pragma unroll
      for (int ksi = 0; ksi < KSW; ++ksi) {
        if (ksi + 1 < KSW) frags(kq + (ksi + 1) * WK, pb, qb, af[(ksi + 1) & 1], bf[(ksi + 1) & 1], i);
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ksi & 1][r], bf[ksi & 1][t], acc[r][t], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (late && iu < L) issue_next();
    cpm += d.SA;
    cpm = cpm >= g.R ? cpm - g.R : cpm;
    ps_c = ps_c + 1 == g.PD ? 0 : ps_c + 1;
    advance(it_c);
  }

  // ---- accumulators -> this workgroup's workspace slot (register order; the index map of the reduction knows it)
  float* wsp = d.ws + ((size_t)((size_t)split * g.DT + dt) * 8 + wave) * (RW * NT * 16 * 64) + lane;
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) wsp[((r * NT + t) * 16 + e) * 64] = acc[r][t][e];
}

static int clw_geometry(const rfx_cl_wgrad_desc& d, ClWgK& k) {
  if (d.N <= 0 || d.OA <= 0 || d.IA <= 0 || d.B <= 0 || (d.PW != 64 && d.PW != 128) || d.B % d.PW) return -1;
  if (d.NTR < 1 || d.NTR > 16 || d.NTC < 1 || d.NTC > 9 || d.SA < 1 || d.SA > d.NTR) return -1;
  if (d.M <= 0 || d.Cq <= 0 || d.CW < 16 || d.CW % 16 || d.CW > 128) return -1;
  if ((d.RW != 2 && d.RW != 3) || (d.WK != 1 && d.WK != 2 && d.WK != 4)) return -1;
  if (d.PW == 64 && d.WK == 4 && false) return -1;
  if (d.p.bs % 8 || d.p.c0 % 8 || d.q.bs % 8 || d.q.c0 % 8 || d.S < 1 || d.ahead < 1) return -1;
  int hb = 0;
  for (int t = 0; t < d.NTC; ++t) {
    const int v = d.db0 + t * d.db_step;
    hb = v < 0 ? (-v > hb ? -v : hb) : (v > hb ? v : hb);
  }
  if (hb > 8) return -1;
  k.d = d;
  k.HB = hb;
  k.nbq = d.B / d.PW;
  k.MTn = (d.M + 32 * d.RW - 1) / (32 * d.RW);
  k.CTn = (d.Cq + d.CW - 1) / d.CW;
  k.DT = k.MTn * k.CTn;
  k.spx = (d.S + 7) / 8;
  const int64_t total = (int64_t)d.N * k.nbq * d.OA;
  if (total >= 0x7fffffffLL) return -1;
  k.total = (int)total;
  k.sps = (int)((total + d.S - 1) / d.S);
  if ((total + k.sps - 1) / k.sps != d.S) return -1;           // every split must own at least one step (the caller rounds S)
  k.NCG = d.CW / 16;
  k.HN = d.NTR * d.NTC * k.NCG;
  const int WC = 8 / d.WK, tiles = (k.HN + 1) / 2;
  if (tiles > WC * 2) return -1;
  k.bias_tile = d.bias ? tiles : -1;
  if (d.bias && tiles + 1 > WC * 2) return -1;
  k.NP = (d.PW * 64 * d.RW + 1023) / 1024;
  k.QROWP = d.CW * 2;
  k.NQ = ((d.PW + 2 * hb) * k.QROWP + 1023) / 1024;
  k.TP = k.NP + d.SA * k.NQ;
  k.PPW = (k.TP + 7) / 8;
  if (k.PPW > CL_WG_MAXP) return -1;
  k.PSLOT = k.NP * 1024;
  // the last transposing read of a tile reaches (HB + db + 63) positions in: the row slot covers NQ KiB >= that by construction
  k.QROWB = k.NQ * 1024;
  k.PRE = (d.NTR - d.SA + d.SA - 1) / d.SA;
  k.PD = d.ahead + 1;
  k.R = ((d.NTR + d.ahead * d.SA + d.SA - 1) / d.SA) * d.SA;
  if ((k.PD - 2) * k.PPW > 24 && k.PD > 2) return -1;
  const int64_t pb = ((int64_t)(d.OA - 1) * d.p.as + (int64_t)d.B * d.p.bs) * 2;
  const int64_t qb = ((int64_t)(d.IA - 1) * d.q.as + (int64_t)d.B * d.q.bs) * 2;
  if (pb <= 0 || qb <= 0 || pb >= 0x7fffffffLL || qb >= 0x7fffffffLL) return -1;
  k.p_bytes = (uint32_t)pb;
  k.q_bytes = (uint32_t)qb;
  return 0;
}

template <int RW, int WK, int PW>
static int clw_launch1(const ClWgK& k, int lds, hipStream_t s) {
  static int attr_lds = 0;
  if (lds > attr_lds) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&cl_wgrad_kernel<RW, WK, PW>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
        hipSuccess)
      return -3;
    attr_lds = lds;
  }
  hipLaunchKernelGGL((cl_wgrad_kernel<RW, WK, PW>), dim3((unsigned)(k.d.S >= 8 ? 8 * k.spx * k.DT : k.d.S * k.DT)), dim3(512), lds, s, k);
  RFX_CHECK_LAUNCH();
  return 0;
}
template <int RW, int WK>
static int clw_launch(const ClWgK& k, int lds, hipStream_t s) {
  return k.d.PW == 128 ? clw_launch1<RW, WK, 128>(k, lds, s) : clw_launch1<RW, WK, 64>(k, lds, s);
}

extern "C" int64_t rfx_cl_wgrad_ws_floats(const rfx_cl_wgrad_desc* dp) {
  ClWgK k;
  if (!dp || clw_geometry(*dp, k)) return -1;
  return (int64_t)dp->S * k.DT * 8 * dp->RW * 2 * 1024;
}

extern "C" int rfx_cl_wgrad(const rfx_cl_wgrad_desc* dp, void* stream) {
  if (!dp || !dp->p.p || !dp->q.p || !dp->ws) return -1;
  ClWgK k;
  if (clw_geometry(*dp, k)) return -1;
  const int lds = k.PD * k.PSLOT + k.R * k.QROWB;
  if (lds > 160 * 1024) return -1;
  hipStream_t s = (hipStream_t)stream;
  if (dp->RW == 3) {
    if (dp->WK == 1) return clw_launch<3, 1>(k, lds, s);
    if (dp->WK == 2) return clw_launch<3, 2>(k, lds, s);
    return clw_launch<3, 4>(k, lds, s);
  }
  if (dp->WK == 1) return clw_launch<2, 1>(k, lds, s);
  if (dp->WK == 2) return clw_launch<2, 2>(k, lds, s);
  return clw_launch<2, 4>(k, lds, s);
}

// ---- fixed-order reduction + scatter ----------------------------------------------------------------------------------------------
// map: one int32 per accumulator element of the kq == 0 waves of every D tile ([DT][WC][RW * 2 * 16][64]): flat index into the weight
// gradient, wn + m for the bias gradient, -1 for cells outside the layer.  Sources are summed split by split, K wave by K wave.
// A block = 64 consecutive map cells x 4 groups of splits: each thread adds its splits in order (8 loads in flight), the four
// group sums are added in group order -- the same tree whatever the launch, so the result is reproducible bit for bit.
__global__ __launch_bounds__(256) void cl_wgrad_reduce_kernel(const float* __restrict__ ws, const int32_t* __restrict__ map, int64_t nmap,
                                                              int S, int DT, int WC, int WK, int per_wave, float* __restrict__ dw,
                                                              int64_t wn, float* __restrict__ db, int accumulate) {
  __shared__ float part[4][64];
  const int l = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const int64_t e = (int64_t)blockIdx.x * 64 + l;
  const int32_t dst = e < nmap ? map[e] : -1;
  float sum = 0.f;
  if (dst >= 0) {
    const int64_t inner = e % per_wave;
    const int64_t dc = e / per_wave;                             // dt * WC + cw
    const int cw = (int)(dc % WC);
    const int64_t dt = dc / WC;
    const int nsrc = S * WK;                                     // source i = (split i / WK, K wave i % WK)
    const int per = (nsrc + 3) / 4;
    const int i0 = sg * per, i1 = min(i0 + per, nsrc);
    const int64_t sstride = (int64_t)DT * 8 * per_wave;
    const float* base = ws + dt * 8 * per_wave + (int64_t)cw * per_wave + inner;
    auto src = [&](int i) { return base[(int64_t)(i / WK) * sstride + (int64_t)(i % WK) * WC * per_wave]; };
    int i = i0;
    for (; i + 8 <= i1; i += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src(i + u);
#pragma unroll
      for (int u = 0; u < 8; ++u) sum += v[u];
    }
    for (; i < i1; ++i) sum += src(i);
  }
  part[sg][l] = sum;
  __syncthreads();
  if (sg == 0 && dst >= 0) {
    const float tot = ((part[0][l] + part[1][l]) + part[2][l]) + part[3][l];
    float* o = dst >= wn ? db + (dst - wn) : dw + dst;
    *o = accumulate ? *o + tot : tot;
  }
}

extern "C" int rfx_cl_wgrad_reduce(const float* ws, const int32_t* map, int64_t nmap, int32_t S, int32_t DT, int32_t RW, int32_t WK,
                                   float* dw, int64_t wn, float* db, int32_t accumulate, void* stream) {
  if (!ws || !map || !dw || nmap <= 0 || S < 1 || DT < 1 || (RW != 2 && RW != 3) || (WK != 1 && WK != 2 && WK != 4)) return -1;
  const int WC = 8 / WK, per_wave = RW * 2 * 16 * 64;
  if (nmap != (int64_t)DT * WC * per_wave) return -1;
  hipLaunchKernelGGL(cl_wgrad_reduce_kernel, dim3((unsigned)((nmap + 63) / 64)), dim3(256), 0, (hipStream_t)stream, ws, map, nmap, S,
                     DT, WC, WK, per_wave, dw, wn, db, accumulate);
  RFX_CHECK_LAUNCH();
  return 0;
}
