// Weight-gradient kernels on the bf16 matrix pipe (shared by gemm_wgrad_bf3.hip / gemm_wgrad_bf16.hip).
#pragma once
#include "gemm_tap.h"

struct WgradArgs {
  rfx_gemm_desc d;
  const rfx_ktab_entry* ktab;
  const float* in;
  const float* g;
  float* dapack;
  int tiles_per_sample;  // ceil(P / 32)
  int total_tiles;       // N * tiles_per_sample
  int tiles_per_block;
  int kt, mt, splits;
  int64_t split_stride;  // floats between the result slices of two position splits (fixed-order reduction in the unpack kernels: no atomics)
  int xcd_grouped;       // 1: 1-D grid, all (k, m) tiles of one position split share an XCD (ids congruent mod 8)
  uint32_t in_bytes, g_bytes;   // wide kernel: exact span of one sample of each operand (buffer num_records)
};

// bf16x3 weight gradient: same tiling and gathers as gemm_wgrad_kernel, but the two LDS tiles hold
// the operands pre-split into bf16 hi / lo halves ([row][32 positions], 80-byte rows: 16-byte aligned
// MFMA fragments, conflict-free ds_read_b128) and the product runs on v_mfma_f32_32x32x16_bf16.
// WM = waves along M: 2 -> the 4 waves form a 2 x 2 grid over a (64 TM) x (64 TK) tile; 1 (M <= 32, the DConv
// bottleneck convs with 12 / 24 output channels) -> 1 x 4 over 32 x (128 TK), so the MFMA rows beyond M and the
// re-loads of g by every k tile are halved.
// MODE 1: split bf16x3 (hi + lo tiles, 3 MFMAs per product); MODE 2: operands rounded to bf16 (hi tile only, 1 MFMA).
// G16: the gradient operand g is stored as bf16 (bf16 mode; strided plans, i.e. the time branch's encoder convs): two-byte loads,
// the stored bits go to LDS as they are.
template <int TM, int TK, int WM, int MODE, bool G16 = false>
__global__ __launch_bounds__(256) void gemm_wgrad_bf_kernel(const WgradArgs w) {
  constexpr int WK = 4 / WM;
  constexpr int RM = 32 * WM * TM, RK = 32 * WK * TK, LDW = 40;   // bf16 elements per LDS row
  constexpr int LO = MODE == 1 ? 1 : 0;                            // no lo tiles in bf16 mode
  __shared__ __attribute__((aligned(16))) unsigned short gs_hi[RM * LDW], gs_lo[LO ? RM * LDW : 8];
  __shared__ __attribute__((aligned(16))) unsigned short xs_hi[RK * LDW], xs_lo[LO ? RK * LDW : 8];
  __shared__ rfx_ktab_entry kts[RK];
  const rfx_gemm_desc& d = w.d;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wm = WM == 2 ? wave >> 1 : 0, wk = WM == 2 ? wave & 1 : wave;
  // plain order (k tile fastest).  An XCD-grouped order (all tiles of one position split on one XCD) was
  // measured SLOWER (132 -> 116 TF/s-eq at the TCN shape): the splits are too few / too coarse to balance.
  int zsplit = blockIdx.z, ym = blockIdx.y, xk = blockIdx.x;
  if (w.xcd_grouped) {
    // every (k, m) tile of a position split re-reads the same g rows / input samples: keep them behind ONE L2
    const int nb = w.kt * w.mt, q = blockIdx.x >> 3;
    zsplit = (q / nb) * 8 + (blockIdx.x & 7);
    if (zsplit >= w.splits) return;
    const int r = q % nb;
    ym = r / w.kt;
    xk = r - ym * w.kt;
  }
  const int m0 = ym * RM;
  const int k0 = xk * RK;
  const int P = d.OA * d.OB;
  for (int i = tid; i < RK; i += 256) {
    rfx_ktab_entry e;
    if (k0 + i < d.Kpad) e = w.ktab[k0 + i];
    else { e.off = 0; e.da = -(1 << 30); e.db = 0; e.flags = 0; }
    kts[i] = e;
  }
  __syncthreads();
  f32x16 acc[TM][TK];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TK; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int t_begin = zsplit * w.tiles_per_block;
  const int t_end = min(t_begin + w.tiles_per_block, w.total_tiles);
  const int prow = tid >> 5, pl = tid & 31;
  // operands of one position tile in flight: raw buffer loads relative to the sample bases (an out-of-range offset
  // reads 0 in hardware: no pointer selects, no branches); the bias ("ones") row is added at staging time
  struct Stage { float gv[RM / 8], xv[RK / 8]; float jv; };
  float onesf[RK / 8];
#pragma unroll
  for (int i = 0; i < RK / 8; ++i) onesf[i] = (kts[prow + 8 * i].flags & 1) ? 1.f : 0.f;
  // tap range of this k tile (wave-uniform): a position tile whose every lane has every tap inside the input skips the two range
  // tests and the select per row; rows that never load (bias row, channel padding) are re-pointed at 2^31 in the LDS copy of the
  // table, which the buffer range check turns into 0.  (Offsets in registers instead -- as in the wide kernel -- cost a resident
  // wave here: DCUNet step 124.6 -> 132.3 ms.)
  int damin = 0, damax = 0, dbmin = 0, dbmax = 0;
  for (int i = (tid & 63); i < RK; i += 64) {
    const rfx_ktab_entry e = kts[i];
    if (!(e.flags & 1) && e.da > -(1 << 29)) {
      damin = min(damin, e.da); damax = max(damax, e.da);
      dbmin = min(dbmin, e.db); dbmax = max(dbmax, e.db);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    damin = min(damin, __shfl_xor(damin, o, 64)); damax = max(damax, __shfl_xor(damax, o, 64));
    dbmin = min(dbmin, __shfl_xor(dbmin, o, 64)); dbmax = max(dbmax, __shfl_xor(dbmax, o, 64));
  }
  damin = __builtin_amdgcn_readfirstlane(damin); damax = __builtin_amdgcn_readfirstlane(damax);
  dbmin = __builtin_amdgcn_readfirstlane(dbmin); dbmax = __builtin_amdgcn_readfirstlane(dbmax);
  __shared__ uint32_t xoffs[RK];
  for (int i = tid; i < RK; i += 256) {
    const rfx_ktab_entry e = kts[i];
    xoffs[i] = ((e.flags & 1) || e.da <= -(1 << 29)) ? RFX_BUF_OOB : ((uint32_t)e.off << 2);
  }
  __syncthreads();
  auto load_tile = [&](int t, Stage& st) {
    const int n = t / w.tiles_per_sample;                       // wave-uniform
    const int j = (t - n * w.tiles_per_sample) * 32 + pl;
    const bool jvalid = j < P;
    const int jj = jvalid ? j : 0;
    const int a = jj / d.OB, b = jj - a * d.OB;
    const int ia0 = a * d.SA, ib0 = b * d.SB;
    const __amdgpu_buffer_rsrc_t irs = rfx_sample_rsrc(w.in + (int64_t)n * d.in_ns);
    constexpr int GSZ = G16 ? 2 : 4;
    const __amdgpu_buffer_rsrc_t grs = rfx_sample_rsrc(reinterpret_cast<const float*>(
        reinterpret_cast<const char*>(w.g) + (int64_t)n * d.out_ns * GSZ));
    const uint32_t voff = (uint32_t)(((int64_t)ia0 * d.in_as + (int64_t)ib0 * d.in_bs) * 4);
    const uint32_t goff = (uint32_t)(((int64_t)(a * d.out_sa + d.out_a0) * d.out_as +
                                      (int64_t)(b * d.out_sb + d.out_b0) * d.out_bs + (int64_t)(m0 + prow) * d.out_cs) * GSZ);
    const uint32_t gstep = (uint32_t)(8 * d.out_cs * GSZ);
    st.jv = jvalid ? 1.f : 0.f;
    const bool inside = jvalid & (ia0 + damin >= 0) & (ia0 + damax < d.IA) & (ib0 + dbmin >= 0) & (ib0 + dbmax < d.IB);
    if (__builtin_amdgcn_ballot_w64(inside) == ~0ull) {          // every lane of the wave: interior position
#pragma unroll
      for (int i = 0; i < RM / 8; ++i) {
        const uint32_t off = (m0 + prow + 8 * i < d.M) ? goff + i * gstep : RFX_BUF_OOB;
        if (G16)
          st.gv[i] = __uint_as_float((uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(grs, off, 0, 0));
        else
          st.gv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(grs, off, 0, 0));
      }
#pragma unroll
      for (int i = 0; i < RK / 8; ++i)
        st.xv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(irs, voff + xoffs[prow + 8 * i], 0, 0));
      return;
    }
#pragma unroll
    for (int i = 0; i < RM / 8; ++i) {
      const bool ok = jvalid & (m0 + prow + 8 * i < d.M);
      if (G16)       // the 16 stored bits, zero-extended (staged without conversion)
        st.gv[i] = __uint_as_float((uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(grs, ok ? goff + i * gstep : RFX_BUF_OOB, 0, 0));
      else
        st.gv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(grs, ok ? goff + i * gstep : RFX_BUF_OOB, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < RK / 8; ++i) {
      const rfx_ktab_entry e = kts[prow + 8 * i];
      const bool ok = jvalid & !(e.flags & 1) & ((unsigned)(ia0 + e.da) < (unsigned)d.IA) &
                      ((unsigned)(ib0 + e.db) < (unsigned)d.IB);     // the bias row loads nothing: it is onesf * jv
      st.xv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(irs, ok ? voff + ((uint32_t)e.off << 2) : RFX_BUF_OOB, 0, 0));
    }
  };
  auto put = [&](unsigned short* hi, unsigned short* lo, int row, float v) {
    const __bf16 h = (__bf16)v;                                   // v_cvt_pk_bf16_f32 (RNE)
    const unsigned short hb = __builtin_bit_cast(unsigned short, h);
    hi[row * LDW + pl] = hb;
    if (MODE == 1) {
      const __bf16 l = (__bf16)(v - __uint_as_float((uint32_t)hb << 16));
      lo[row * LDW + pl] = __builtin_bit_cast(unsigned short, l);
    }
  };
  auto stage = [&](const Stage& st) {
#pragma unroll
    for (int i = 0; i < RM / 8; ++i) {
      if (G16) gs_hi[(prow + 8 * i) * LDW + pl] = (unsigned short)__float_as_uint(st.gv[i]);
      else put(gs_hi, gs_lo, prow + 8 * i, st.gv[i]);
    }
#pragma unroll
    for (int i = 0; i < RK / 8; ++i) put(xs_hi, xs_lo, prow + 8 * i, st.xv[i] + onesf[i] * st.jv);
  };
  auto mma_tile = [&]() {
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2) {
      bf16x8 ah[TM], al[TM], bh[TK], bl[TK];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int off = (wm * 32 * TM + tm * 32 + l31) * LDW + 16 * ks2 + 8 * h;
        ah[tm] = *reinterpret_cast<const bf16x8*>(gs_hi + off);
        if (MODE == 1) al[tm] = *reinterpret_cast<const bf16x8*>(gs_lo + off);
      }
#pragma unroll
      for (int tk = 0; tk < TK; ++tk) {
        const int off = (wk * 32 * TK + tk * 32 + l31) * LDW + 16 * ks2 + 8 * h;
        bh[tk] = *reinterpret_cast<const bf16x8*>(xs_hi + off);
        if (MODE == 1) bl[tk] = *reinterpret_cast<const bf16x8*>(xs_lo + off);
      }
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tk = 0; tk < TK; ++tk) {
          acc[tm][tk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bh[tk], acc[tm][tk], 0, 0, 0);
          if (MODE == 1) {
            acc[tm][tk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bl[tk], acc[tm][tk], 0, 0, 0);
            acc[tm][tk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[tm], bh[tk], acc[tm][tk], 0, 0, 0);
          }
        }
    }
  };
  // two tiles of operands in flight: the loads of tile t+2 are issued before the MFMAs of tile t, so an HBM round
  // trip (~1-2 us) is covered by two tiles of matrix work instead of one (the r01 version waited at every tile)
  Stage s0, s1;
  const int t_last = t_end - 1;
  if (t_begin < t_end) {
    load_tile(t_begin, s0);
    load_tile(min(t_begin + 1, t_last), s1);
  }
  for (int t = t_begin; t < t_end; t += 2) {
    __syncthreads();
    stage(s0);
    __syncthreads();
    load_tile(min(t + 2, t_last), s0);
    mma_tile();
    if (t + 1 < t_end) {                       // block-uniform
      __syncthreads();
      stage(s1);
      __syncthreads();
      load_tile(min(t + 3, t_last), s1);
      mma_tile();
    }
  }
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tk = 0; tk < TK; ++tk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 * TM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int k = k0 + wk * 32 * TK + tk * 32 + l31;
        if (m < d.M && k < d.K) w.dapack[(int64_t)zsplit * w.split_stride + (int64_t)m * d.Kpad + k] = acc[tm][tk][r];
      }
}


// ---------------------------------------------------------------------------------
// Wide-load weight gradient (the main path for unit-stride layers: in_bs == out_bs == 1, SB == out_sb == 1).
//   dapack[m][k] += sum_p g[m][p] * In(k, p)
// Both operands are contiguous along the reduction axis p, so a thread moves FOUR consecutive positions of one row per
// instruction (raw buffer dwordx4 = 16 B per lane; a wave instruction covers 4 rows x 256 contiguous bytes), converts
// them with two v_cvt_pk_bf16_f32 and stages them with ONE ds_write_b64 -- the 32-position kernel above issues a
// dword load, two converts and two ds_write_b16 per ELEMENT and was bound by those (PMC r01: MFMA busy 18 %).
// Position chunk = 64 consecutive b positions of one (n, a) output row; LDS rows are 72 bf16 (144 B = 36 dwords: the 16
// lanes of a ds_read_b128 group land on 16 distinct 4-bank groups); 4 MFMA K steps per chunk and per wave tile.
// Borders: a quad that starts left of its row (tap shift db < 0 at b = 0) takes four masked dword loads instead (a
// negative offset would fail the whole dwordx4 range check, scripts/probes/bufprobe.hip); elements right of the row
// end / beyond OB are zeroed after the load; rows whose a-coordinate is out of range load nothing (offset 2^31).
// ---------------------------------------------------------------------------------
template <int TM, int TK, int WM, int MODE, bool G16 = false>     // G16: the gradient operand g is stored as bf16 (bf16 mode only)
// (four workgroups per CU for the <= 4-tile shapes spills inside the chunk loop: 128 x 128 on 192 -> 384 3x3 1.58 -> 2.36 ms, step +10 ms)
__global__ __launch_bounds__(256, (MODE == 2 && TM * TK <= 4) ? 3 : 2) void gemm_wgrad_wide_kernel(const WgradArgs w) {
  constexpr int WK = 4 / WM;
  constexpr int RM = 32 * WM * TM, RK = 32 * WK * TK, LDW = 72, PC = 64;
  constexpr int LO = MODE == 1 ? 1 : 0;
  constexpr int NG = RM / 16, NX = RK / 16;                         // quad loads per thread and chunk
  __shared__ __attribute__((aligned(16))) unsigned short gs_hi[RM * LDW], gs_lo[LO ? RM * LDW : 8];
  __shared__ __attribute__((aligned(16))) unsigned short xs_hi[RK * LDW], xs_lo[LO ? RK * LDW : 8];
  __shared__ rfx_ktab_entry kts[RK];
  const rfx_gemm_desc& d = w.d;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wm = WM == 1 ? 0 : wave / WK, wk = WM == 1 ? wave : wave % WK;
  int zsplit = blockIdx.z, ym = blockIdx.y, xk = blockIdx.x;
  if (w.xcd_grouped) {
    // every (k, m) tile of a position split re-reads the same g rows / input samples: keep them behind ONE L2
    const int nb = w.kt * w.mt, q = blockIdx.x >> 3;
    zsplit = (q / nb) * 8 + (blockIdx.x & 7);
    if (zsplit >= w.splits) return;
    const int r = q % nb;
    ym = r / w.kt;
    xk = r - ym * w.kt;
  }
  const int m0 = ym * RM, k0 = xk * RK;
  for (int i = tid; i < RK; i += 256) {
    rfx_ktab_entry e;
    if (k0 + i < d.Kpad) e = w.ktab[k0 + i];
    else { e.off = 0; e.da = -(1 << 30); e.db = 0; e.flags = 0; }
    kts[i] = e;
  }
  __syncthreads();
  f32x16 acc[TM][TK];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TK; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int t_begin = zsplit * w.tiles_per_block;
  const int t_end = min(t_begin + w.tiles_per_block, w.total_tiles);
  // wave-uniform facts about this k tile's table rows: the b-shift range of its taps and whether it holds the bias ("ones") row.  A
  // chunk whose every quad stays inside its row for every tap takes the lean load path below.
  int dbmin = 0, dbmax = 0, damin = 0, damax = 0, any_ones = 0;
  for (int i = lane; i < RK; i += 64) {
    const rfx_ktab_entry e = kts[i];
    if (e.flags & 1) any_ones = 1;
    else if (e.da > -(1 << 29)) {
      dbmin = min(dbmin, e.db); dbmax = max(dbmax, e.db);
      damin = min(damin, e.da); damax = max(damax, e.da);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    dbmin = min(dbmin, __shfl_xor(dbmin, o, 64)); dbmax = max(dbmax, __shfl_xor(dbmax, o, 64));
    damin = min(damin, __shfl_xor(damin, o, 64)); damax = max(damax, __shfl_xor(damax, o, 64));
    any_ones |= __shfl_xor(any_ones, o, 64);
  }
  dbmin = __builtin_amdgcn_readfirstlane(dbmin); dbmax = __builtin_amdgcn_readfirstlane(dbmax);
  damin = __builtin_amdgcn_readfirstlane(damin); damax = __builtin_amdgcn_readfirstlane(damax);
  any_ones = __builtin_amdgcn_readfirstlane(any_ones);
  const int q4 = (tid & 15) * 4, r0 = tid >> 4;                    // this thread's quad inside the chunk / first row
  const int chunks_per_row = (d.OB + PC - 1) / PC;
  // Per-thread byte offsets of this thread's NX table rows, taken from the table ONCE: rows that never load (bias row, channel
  // padding) get 2^31, which pushes any sum with a sample-relative offset (< 2^31) beyond num_records -> 0.  The chunk loop of the
  // r03 kernel re-read the table from LDS and re-derived offset, row test and select for every row of every chunk: ablation builds
  //  showed the 3x3 layers taking the same 2.2 ms with the load instructions REMOVED -- the kernel was bound by that
  // per-chunk address arithmetic, not by the loads.
  uint32_t xoff[NX];
  uint32_t onesmask = 0;
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const rfx_ktab_entry e = kts[r0 + 16 * i];
    const bool ones = e.flags & 1;
    xoff[i] = (ones || e.da <= -(1 << 29)) ? RFX_BUF_OOB : ((uint32_t)e.off << 2);
    if (ones) onesmask |= 1u << i;
  }
  struct Stage { f32x4 gv[NG], xv[NX]; };
  auto load_chunk = [&](int n, int a, int c0, Stage& st, bool valid) {   // (n, a, c0) wave-uniform; !valid: every load gets the out-of-range offset -> zeros
    const int ob = c0 + q4;                                        // first position of this thread's quad
    if (valid && c0 + PC <= d.OB && c0 + dbmin >= 0 && c0 + PC + dbmax <= d.IB && a * d.SA + damin >= 0 &&
        a * d.SA + damax < d.IA) {
      // FULLY interior chunk (wave-uniform test): every tap of every row is inside the input on both axes -> one add per load
      const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w.in + (int64_t)n * d.in_ns), 0,
                                                                           (int)w.in_bytes, 0x00020000);
      constexpr int GSZ = G16 ? 2 : 4;
      const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(reinterpret_cast<const char*>(w.g) + (int64_t)n * d.out_ns * GSZ), 0, (int)w.g_bytes, 0x00020000);
      const uint32_t goff = (uint32_t)(((int64_t)(a * d.out_sa + d.out_a0) * d.out_as + (int64_t)(ob + d.out_b0) +
                                        (int64_t)(m0 + r0) * d.out_cs) * GSZ);
      const uint32_t gstep = (uint32_t)(16 * d.out_cs * GSZ);
#pragma unroll
      for (int i = 0; i < NG; ++i) {
        const uint32_t off = (m0 + r0 + 16 * i < d.M) ? goff + i * gstep : RFX_BUF_OOB;
        if (G16) {
          const uint2 u = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(grs, off, 0, 0));
          st.gv[i] = f32x4{__uint_as_float(u.x), __uint_as_float(u.y), 0.f, 0.f};
        } else {
          st.gv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(grs, off, 0, 0));
        }
      }
      const uint32_t voff = (uint32_t)(((int64_t)a * d.SA * d.in_as + (int64_t)ob) * 4);
#pragma unroll
      for (int i = 0; i < NX; ++i)
        st.xv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(irs, voff + xoff[i], 0, 0));
      if (any_ones) {                                              // the tile that holds the bias-gradient column (wave-uniform)
#pragma unroll
        for (int i = 0; i < NX; ++i)
          if (onesmask & (1u << i)) st.xv[i] = f32x4{1.f, 1.f, 1.f, 1.f};
      }
      return;
    }
    if (valid && c0 + PC <= d.OB && c0 + dbmin >= 0 && c0 + PC + dbmax <= d.IB) {
      // INTERIOR chunk (wave-uniform test): every quad of every tap lies inside its row -- no edge selects, no straddle loads.  The
      // general path below spends ~330 VALU instructions per chunk on them against 16 MFMAs (r03 ISA count): the kernel was VALU-bound.
      const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w.in + (int64_t)n * d.in_ns), 0,
                                                                           (int)w.in_bytes, 0x00020000);
      constexpr int GSZ = G16 ? 2 : 4;
      const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(reinterpret_cast<const char*>(w.g) + (int64_t)n * d.out_ns * GSZ), 0, (int)w.g_bytes, 0x00020000);
      const uint32_t goff = (uint32_t)(((int64_t)(a * d.out_sa + d.out_a0) * d.out_as + (int64_t)(ob + d.out_b0) +
                                        (int64_t)(m0 + r0) * d.out_cs) * GSZ);
      const uint32_t gstep = (uint32_t)(16 * d.out_cs * GSZ);
#pragma unroll
      for (int i = 0; i < NG; ++i) {
        const uint32_t off = (m0 + r0 + 16 * i < d.M) ? goff + i * gstep : RFX_BUF_OOB;
        if (G16) {
          const uint2 u = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(grs, off, 0, 0));
          st.gv[i] = f32x4{__uint_as_float(u.x), __uint_as_float(u.y), 0.f, 0.f};
        } else {
          st.gv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(grs, off, 0, 0));
        }
      }
      const int ia0 = a * d.SA;
      const uint32_t voff = (uint32_t)(((int64_t)ia0 * d.in_as + (int64_t)ob) * 4);
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const rfx_ktab_entry e = kts[r0 + 16 * i];
        const bool rowok = !(e.flags & 1) & ((unsigned)(ia0 + e.da) < (unsigned)d.IA);
        f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(irs, rowok ? voff + ((uint32_t)e.off << 2) : RFX_BUF_OOB, 0, 0));
        if (any_ones && (e.flags & 1)) v = f32x4{1.f, 1.f, 1.f, 1.f};          // bias-gradient column (any_ones: wave-uniform)
        st.xv[i] = v;
      }
      return;
    }
    const int lim = valid ? d.OB - ob : 0;                         // valid elements of the quad (<= 0: none)
    // num_records = the sample's exact span: a quad that runs past the last row of the tensor reads 0 for the dwords
    // beyond it (per-dword range check) instead of touching memory behind the allocation
    const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w.in + (int64_t)n * d.in_ns), 0,
                                                                         (int)w.in_bytes, 0x00020000);
    constexpr int GSZ = G16 ? 2 : 4;
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(w.g) + (int64_t)n * d.out_ns * GSZ), 0, (int)w.g_bytes, 0x00020000);
    const uint32_t goff = (uint32_t)(((int64_t)(a * d.out_sa + d.out_a0) * d.out_as + (int64_t)(ob + d.out_b0) +
                                      (int64_t)(m0 + r0) * d.out_cs) * GSZ);
    const uint32_t gstep = (uint32_t)(16 * d.out_cs * GSZ);
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const bool ok = (lim > 0) & (m0 + r0 + 16 * i < d.M);
      if (G16) {                                                   // 4 positions = 8 bytes, already bf16: no conversion at staging
        uint2 u = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(grs, ok ? goff + i * gstep : RFX_BUF_OOB, 0, 0));
        u.x = lim > 1 ? u.x : (u.x & 0xffffu);                     // a row end inside the quad
        u.y = lim > 3 ? u.y : (lim > 2 ? (u.y & 0xffffu) : 0u);
        st.gv[i] = f32x4{__uint_as_float(u.x), __uint_as_float(u.y), 0.f, 0.f};
      } else {
        f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(grs, ok ? goff + i * gstep : RFX_BUF_OOB, 0, 0));
#pragma unroll
        for (int e = 1; e < 4; ++e) v[e] = e < lim ? v[e] : 0.f;   // a row end inside the quad (OA == 1, OB % 4 != 0)
        st.gv[i] = v;
      }
    }
    const int ia0 = a * d.SA;
    const uint32_t voff = (uint32_t)(((int64_t)ia0 * d.in_as + (int64_t)ob) * 4);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const rfx_ktab_entry e = kts[r0 + 16 * i];
      const bool ones = e.flags & 1;
      const bool rowok = (lim > 0) & !ones & ((unsigned)(ia0 + e.da) < (unsigned)d.IA);
      const int ib0 = ob + e.db;                                   // SB == 1
      const int hi = min(lim, d.IB - ib0);                         // elements [0, hi) are inside the row on the right
      const uint32_t off = voff + ((uint32_t)e.off << 2);
      f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(irs, (rowok & (ib0 >= 0)) ? off : RFX_BUF_OOB, 0, 0));
      if (rowok & (ib0 < 0) & (ib0 > -4)) {                        // rare: the quad straddles the left end of its row
#pragma unroll
        for (int q = 1; q < 4; ++q)
          v[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(irs, (ib0 + q >= 0) ? off + 4 * q : RFX_BUF_OOB, 0, 0));
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = q < hi ? v[q] : 0.f;
      if (ones) {                                                  // bias-gradient column: 1 at every valid position
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = q < lim ? 1.f : 0.f;
      }
      st.xv[i] = v;
    }
  };
  auto put4 = [&](unsigned short* hi, unsigned short* lo, int row, const f32x4& v) {
    const f32x2_t a = {v[0], v[1]}, b = {v[2], v[3]};
    const uint32_t ha = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, bf16x2_t));
    const uint32_t hb = __builtin_bit_cast(uint32_t, __builtin_convertvector(b, bf16x2_t));
    *reinterpret_cast<uint2*>(hi + row * LDW + q4) = make_uint2(ha, hb);
    if (MODE == 1) {
      const f32x2_t fa = {__uint_as_float(ha << 16), __uint_as_float(ha & 0xffff0000u)};
      const f32x2_t fb = {__uint_as_float(hb << 16), __uint_as_float(hb & 0xffff0000u)};
      const uint32_t la = __builtin_bit_cast(uint32_t, __builtin_convertvector(a - fa, bf16x2_t));
      const uint32_t lb = __builtin_bit_cast(uint32_t, __builtin_convertvector(b - fb, bf16x2_t));
      *reinterpret_cast<uint2*>(lo + row * LDW + q4) = make_uint2(la, lb);
    }
  };
  auto stage = [&](const Stage& st) {
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      if (G16) *reinterpret_cast<uint2*>(gs_hi + (r0 + 16 * i) * LDW + q4) = make_uint2(__float_as_uint(st.gv[i][0]), __float_as_uint(st.gv[i][1]));
      else put4(gs_hi, gs_lo, r0 + 16 * i, st.gv[i]);
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) put4(xs_hi, xs_lo, r0 + 16 * i, st.xv[i]);
  };
  auto mma_chunk = [&]() {
#pragma unroll
    for (int ks = 0; ks < PC / 16; ++ks) {
      bf16x8 ah[TM], al[TM], bh[TK], bl[TK];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int off = (wm * 32 * TM + tm * 32 + l31) * LDW + 16 * ks + 8 * h;
        ah[tm] = *reinterpret_cast<const bf16x8*>(gs_hi + off);
        if (MODE == 1) al[tm] = *reinterpret_cast<const bf16x8*>(gs_lo + off);
      }
#pragma unroll
      for (int tk = 0; tk < TK; ++tk) {
        const int off = (wk * 32 * TK + tk * 32 + l31) * LDW + 16 * ks + 8 * h;
        bh[tk] = *reinterpret_cast<const bf16x8*>(xs_hi + off);
        if (MODE == 1) bl[tk] = *reinterpret_cast<const bf16x8*>(xs_lo + off);
      }
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tk = 0; tk < TK; ++tk) {
          acc[tm][tk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bh[tk], acc[tm][tk], 0, 0, 0);
          if (MODE == 1) {
            acc[tm][tk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bl[tk], acc[tm][tk], 0, 0, 0);
            acc[tm][tk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[tm], bh[tk], acc[tm][tk], 0, 0, 0);
          }
        }
    }
  };
  // Register stages.  With ONE stage the loads of chunk t+1 are in flight only under chunk t's MFMAs (~0.3 us) while an L2 / HBM
  // round trip under load is 1-2 us: the r02b launch list showed the MFMA-heavy layers (3x3 rewrite convs, K = 9 Cin) at 270-360
  // TF/s, i.e. waiting on loads.  Where two stages fit the register budget (everything but the 96 x 256 tile) the loop runs two
  // chunks per iteration and each stage's loads fly under TWO chunks of staging + MFMAs.  An odd tail chunk is loaded with the
  // out-of-range offset (zeros), so the loop body is branch-free.
  // MEASURED (r02b, Demucs step): weight-gradient launches 36.3 -> 39.5 ms with the two-stage loop, the 3x3 layers unchanged
  // (2.56 -> 2.59 ms): they are bound by LDS / L1 bandwidth (48-64 KB into the CU per 512 clk of MFMA work), not by load latency,
  // and the extra registers cost the small tiles occupancy.  Re-measured in r03 on the lean-load kernel (bf16 mode, two workgroups per
  // CU instead of three to make room): 128 x 128 tiles with two stages 1.59 -> 1.85 ms (192 -> 384, 3x3) and 1.94 -> 2.43 ms
  // (384 -> 768), step 145.4 -> 147.4 ms; the 96-row layers on 96 x 128 tiles with two stages instead of 96 x 256 with one 2.22 ->
  // 2.66 ms, step +0.9 ms: the third resident workgroup hides more latency than the second register stage.  Kept for the record, off.
  const int t_last = t_end - 1;
  // position of the chunk being loaded, advanced incrementally (the r03 loop decoded t with two integer divisions per chunk)
  int pn = 0, pa = 0, pc0 = 0;
  if (t_begin < t_end) {
    const int row = t_begin / chunks_per_row;
    pn = row / d.OA; pa = row - pn * d.OA; pc0 = (t_begin - row * chunks_per_row) * PC;
  }
  pn = __builtin_amdgcn_readfirstlane(pn); pa = __builtin_amdgcn_readfirstlane(pa); pc0 = __builtin_amdgcn_readfirstlane(pc0);
  Stage st;
  if (t_begin < t_end) load_chunk(pn, pa, pc0, st, true);
  for (int t = t_begin; t < t_end; ++t) {
    __syncthreads();                       // the previous chunk's fragment reads are done
    stage(st);
    __syncthreads();
    if (t < t_last) {                      // wave-uniform; on the last iteration the same chunk is re-read (branch-free loads)
      pc0 += PC;
      if (pc0 >= chunks_per_row * PC) {
        pc0 = 0;
        if (++pa == d.OA) { pa = 0; ++pn; }
      }
    }
    load_chunk(pn, pa, pc0, st, true);
    mma_chunk();
  }
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tk = 0; tk < TK; ++tk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 * TM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int k = k0 + wk * 32 * TK + tk * 32 + l31;
        if (m < d.M && k < d.K) w.dapack[(int64_t)zsplit * w.split_stride + (int64_t)m * d.Kpad + k] = acc[tm][tk][r];
      }
}

// shape: 0 = 96-row tiles (waves 1 x 4), 1 / 2 = 32-row tiles with 256 / 128 k rows, 3..6 = (64 TM) x (64 TK) tiles
template <int MODE>
static int rfx_launch_wgrad_bf(const WgradArgs& w, int shape, dim3 grid, hipStream_t s) {
  if (w.d.out_bf16) {
    if (MODE != 2) return -1;
    switch (shape) {
      case 0: hipLaunchKernelGGL((gemm_wgrad_bf_kernel<3, 1, 1, 2, true>), grid, dim3(256), 0, s, w); break;
      case 1: hipLaunchKernelGGL((gemm_wgrad_bf_kernel<1, 2, 1, 2, true>), grid, dim3(256), 0, s, w); break;
      case 2: hipLaunchKernelGGL((gemm_wgrad_bf_kernel<1, 1, 1, 2, true>), grid, dim3(256), 0, s, w); break;
      case 3: hipLaunchKernelGGL((gemm_wgrad_bf_kernel<2, 2, 2, 2, true>), grid, dim3(256), 0, s, w); break;
      case 4: hipLaunchKernelGGL((gemm_wgrad_bf_kernel<2, 1, 2, 2, true>), grid, dim3(256), 0, s, w); break;
      case 5: hipLaunchKernelGGL((gemm_wgrad_bf_kernel<1, 2, 2, 2, true>), grid, dim3(256), 0, s, w); break;
      default: hipLaunchKernelGGL((gemm_wgrad_bf_kernel<1, 1, 2, 2, true>), grid, dim3(256), 0, s, w); break;
    }
    RFX_CHECK_LAUNCH();
    return 0;
  }
  switch (shape) {
    case 0: hipLaunchKernelGGL((gemm_wgrad_bf_kernel<3, 1, 1, MODE>), grid, dim3(256), 0, s, w); break;
    case 1: hipLaunchKernelGGL((gemm_wgrad_bf_kernel<1, 2, 1, MODE>), grid, dim3(256), 0, s, w); break;
    case 2: hipLaunchKernelGGL((gemm_wgrad_bf_kernel<1, 1, 1, MODE>), grid, dim3(256), 0, s, w); break;
    case 3: hipLaunchKernelGGL((gemm_wgrad_bf_kernel<2, 2, 2, MODE>), grid, dim3(256), 0, s, w); break;
    case 4: hipLaunchKernelGGL((gemm_wgrad_bf_kernel<2, 1, 2, MODE>), grid, dim3(256), 0, s, w); break;
    case 5: hipLaunchKernelGGL((gemm_wgrad_bf_kernel<1, 2, 2, MODE>), grid, dim3(256), 0, s, w); break;
    default: hipLaunchKernelGGL((gemm_wgrad_bf_kernel<1, 1, 2, MODE>), grid, dim3(256), 0, s, w); break;
  }
  RFX_CHECK_LAUNCH();
  return 0;
}
// wide-load kernel, shape: 0 = 96 x 128 (waves 1 x 4), 1 = 32 x 256, 2 = 32 x 128, 3 = 128 x 128, 4 = 64 x 128, 5 = 96 x 256
template <int MODE>
static int rfx_launch_wgrad_wide(const WgradArgs& w, int shape, dim3 grid, hipStream_t s) {
  if (w.d.out_bf16) {
    if (MODE != 2) return -1;
    switch (shape) {
      case 0: hipLaunchKernelGGL((gemm_wgrad_wide_kernel<3, 1, 1, 2, true>), grid, dim3(256), 0, s, w); break;
      case 1: hipLaunchKernelGGL((gemm_wgrad_wide_kernel<1, 2, 1, 2, true>), grid, dim3(256), 0, s, w); break;
      case 2: hipLaunchKernelGGL((gemm_wgrad_wide_kernel<1, 1, 1, 2, true>), grid, dim3(256), 0, s, w); break;
      case 3: hipLaunchKernelGGL((gemm_wgrad_wide_kernel<2, 2, 2, 2, true>), grid, dim3(256), 0, s, w); break;
      case 5: hipLaunchKernelGGL((gemm_wgrad_wide_kernel<3, 2, 1, 2, true>), grid, dim3(256), 0, s, w); break;
      default: hipLaunchKernelGGL((gemm_wgrad_wide_kernel<1, 2, 2, 2, true>), grid, dim3(256), 0, s, w); break;
    }
    RFX_CHECK_LAUNCH();
    return 0;
  }
  switch (shape) {
    case 0: hipLaunchKernelGGL((gemm_wgrad_wide_kernel<3, 1, 1, MODE>), grid, dim3(256), 0, s, w); break;
    case 1: hipLaunchKernelGGL((gemm_wgrad_wide_kernel<1, 2, 1, MODE>), grid, dim3(256), 0, s, w); break;
    case 2: hipLaunchKernelGGL((gemm_wgrad_wide_kernel<1, 1, 1, MODE>), grid, dim3(256), 0, s, w); break;
    case 3: hipLaunchKernelGGL((gemm_wgrad_wide_kernel<2, 2, 2, MODE>), grid, dim3(256), 0, s, w); break;
    case 5: hipLaunchKernelGGL((gemm_wgrad_wide_kernel<3, 2, 1, MODE>), grid, dim3(256), 0, s, w); break;
    default: hipLaunchKernelGGL((gemm_wgrad_wide_kernel<1, 2, 2, MODE>), grid, dim3(256), 0, s, w); break;
  }
  RFX_CHECK_LAUNCH();
  return 0;
}
int rfx_launch_wgrad_wide_bf3(const WgradArgs& w, int shape, dim3 grid, hipStream_t s);
int rfx_launch_wgrad_wide_bf16(const WgradArgs& w, int shape, dim3 grid, hipStream_t s);
int rfx_launch_wgrad_bf3(const WgradArgs& w, int shape, dim3 grid, hipStream_t s);
int rfx_launch_wgrad_bf16(const WgradArgs& w, int shape, dim3 grid, hipStream_t s);
