// Gather-GEMM family: every convolution on the RemFX hot path (forward, input
// gradient, transposed forward, weight gradient) as one table-driven implicit
// GEMM on the fp32 MFMA pipe of gfx950 (v_mfma_f32_32x32x2_f32, exact fp32).
//
// Layout of the forward kernel (wave64, 4 waves / workgroup):
//   * workgroup tile = (32*R) output channels x 128 output positions; wave w owns
//     positions [32w, 32w+32) and all R channel tiles -> the B operand (gathered
//     input samples) goes global -> VGPR directly, one dword per lane per MFMA
//     k-pair, coalesced along the contiguous position axis; no LDS round trip.
//   * the A operand (packed weights [Kpad][Mpad], M contiguous) is staged in LDS
//     with 16-byte loads, double buffered, one barrier per 16-deep K step.
//   * per-k tap metadata (offset + displacement for the bounds test) is wave
//     uniform and read through the scalar cache.
// Replaces F.conv1d/conv2d/conv_transpose1d/2d call sites: tcn.py:50,54,129;
// HDemucs / DCUNet / Cnn14 stacks (models.py:319,358; classifier.py:271-272).
#include <stdlib.h>
// Shared by the translation units of the family (one per kernel group so that hipcc works on them in parallel):
// gemm.hip (packing, thin forward, C entry points), gemm_fwd_f32.hip / gemm_fwd_bf3.hip (the tiled forward kernel in
// its two arithmetic modes), gemm_wgrad.hip (weight gradient).
#pragma once
#include <stdlib.h>

#include "common.h"

struct FwdArgs {
  rfx_gemm_desc d;
  const float* apack;
  const rfx_ktab_entry* ktab;
  const float* in;
  float* out;
  rfx_epilogue e;
  const float* apack2;
  const rfx_ktab_entry* ktab2;
  int32_t Kpad2;
  const float* in2;
  int32_t ntaps2;     // tap-major launches: taps of the second phase (ktab / ktab2 are tap tables there)
  int32_t pair_store; // 16-bit outputs: two positions per store (bf16_pair_store); RFX_PAIR_STORE=0 switches it off
  int32_t xcd_chunk;  // > 0: XCD x walks the CONTIGUOUS work items [x * chunk, (x + 1) * chunk) (plans whose taps span rows of the
                      // A axis: neighbouring position tiles share input rows and should meet in one L2); 0: round-robin
};


typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t bf16_rne(float f) { return rfx_bf16_bits(f); }

static inline bool desc_ok(const rfx_gemm_desc* d) {
  return d && d->N > 0 && d->M > 0 && d->K >= 0 && d->OA > 0 && d->OB > 0 && d->Mpad % 4 == 0 &&
         d->Kpad % 16 == 0 && d->Kpad >= d->K && d->Mpad >= d->M;
}


// ---------------------------------------------------------------------------------
// forward MFMA kernel
// ---------------------------------------------------------------------------------
struct LaneCtx {
  const float* inb;  // in + n*in_ns + position offset
  const float* safe; // always-valid address
  int ia0, ib0;
  bool jvalid;
  // bf16x3 path: gathers are raw buffer loads relative to the sample base; an out-of-range offset makes the
  // hardware return 0 without touching memory, so masking costs one 32-bit select instead of a 64-bit pointer select
  __amdgpu_buffer_rsrc_t rs;
  uint32_t voff;     // byte offset of this lane's position inside the sample
};
#define RFX_BUF_OOB 0x80000000u       // > num_records (0x7fffffff): reads as 0

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rfx_sample_rsrc(const float* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
}

// A zero the invalid lanes can load instead of masking the loaded value: any VALU op
// on the result forces s_waitcnt vmcnt(0) right after the load and exposes the full
// memory latency every K step (measured: 47 -> see profiles/ after the change).
static __device__ float rfx_zero_f32[4] = {0.f, 0.f, 0.f, 0.f};
static __device__ float rfx_one_f32[4] = {1.f, 1.f, 1.f, 1.f};

// ktl: the 16 tap entries of this K step, staged in LDS (scalar loads of the table
// serialise on lgkmcnt(0) per entry; LDS broadcast reads do not).
__device__ __forceinline__ void load_b8(const rfx_gemm_desc& d, const int4* ktl, int h, const LaneCtx& c,
                                        float (&b)[8]) {
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    int4 e[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) e[q] = ktl[2 * (half * 4 + q) + h];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool ok = c.jvalid & ((unsigned)(c.ia0 + e[q].y) < (unsigned)d.IA) &
                      ((unsigned)(c.ib0 + e[q].z) < (unsigned)d.IB);
      const float* p = ok ? (c.inb + e[q].x) : c.safe;
      b[half * 4 + q] = *p;
    }
  }
}

template <int R>
__device__ __forceinline__ void stage_a_load(const float* __restrict__ apack, int Mpad, int k0, int m0,
                                             int tid, f32x4 (&r)[2]) {
  constexpr int BM = 32 * R;
  constexpr int NV = 16 * BM / 4;  // float4 per tile
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (i * 256 < NV) {            // compile-time
      int idx = tid + i * 256;
      idx = idx < NV ? idx : NV - 1;   // branch-free: surplus threads re-read the last vector
      const int kk = idx / (BM / 4), c4 = idx % (BM / 4);
      r[i] = *reinterpret_cast<const f32x4*>(apack + (int64_t)(k0 + kk) * Mpad + m0 + 4 * c4);
    }
  }
}
template <int R>
__device__ __forceinline__ void stage_a_store(float* as, int tid, const f32x4 (&r)[2]) {
  constexpr int BM = 32 * R;
  constexpr int NV = 16 * BM / 4;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (i * 256 < NV) {            // compile-time
      // UNCONDITIONAL store (surplus threads rewrite the last vector with the same data): a store under a
      // lane condition lets LLVM sink the global load into that branch, right in front of a vmcnt(0)
      int idx = tid + i * 256;
      idx = idx < NV ? idx : NV - 1;
      const int kk = idx / (BM / 4), c4 = idx % (BM / 4);
      *reinterpret_cast<f32x4*>(as + kk * BM + 4 * c4) = r[i];
    }
  }
}

// One 16-deep K step: issue next step's operands (A -> regs, B gathers -> bn), run the
// 8*R MFMAs of this step on (LDS A[cur], bc), publish A[cur^1] / table rows, barrier.
// The body is BRANCH-FREE: the packed A matrix carries one extra all-zero K step and
// the tap table two extra all-invalid steps, so the prefetch of step ks+1 / ks+2 is
// unconditional.  (With `if (more)` around the loads hipcc's waitcnt pass merges the
// two paths and drains vmcnt to 0 in front of the MFMAs: every gather's latency exposed.)
// bc/bn ping-pong between two register sets, so the only vmcnt wait is the counted one
// in front of the NEXT step's MFMAs.
template <int R>
__device__ __forceinline__ void k_step(const rfx_gemm_desc& d, const float* __restrict__ apack,
                                       const int4* __restrict__ kt4, int ks, int m0,
                                       const LaneCtx& c, float* as, int4* kts, f32x16 (&acc)[R],
                                       const float (&bc)[8], float (&bn)[8]) {
  constexpr int BM = 32 * R;
  const int tid = threadIdx.x;
  const int lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int cur = ks & 1;
  const float* a_lds = as + cur * 16 * BM;
  float afrag[8][R];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk)
#pragma unroll
    for (int mt = 0; mt < R; ++mt) afrag[kk][mt] = a_lds[(2 * kk + h) * BM + mt * 32 + l31];
  const int4 ktreg = kt4[(ks + 2) * 16 + (tid & 15)];     // consumed this step: issued before the gathers (in-order vmcnt)
  f32x4 areg[2];
  stage_a_load<R>(apack, d.Mpad, (ks + 1) * 16, m0, tid, areg);
  load_b8(d, kts + ((ks + 1) % 3) * 16, h, c, bn);
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
#pragma unroll
    for (int mt = 0; mt < R; ++mt)
      acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[kk][mt], bc[kk], acc[mt], 0, 0, 0);
  }
  stage_a_store<R>(as + (cur ^ 1) * 16 * BM, tid, areg);
  kts[((ks + 2) % 3) * 16 + (tid & 15)] = ktreg;      // every thread (same value per tid & 15): no lane condition
  __syncthreads();
}

template <int R>
__device__ __forceinline__ void run_phase(const rfx_gemm_desc& d, const float* __restrict__ apack,
                                          const rfx_ktab_entry* __restrict__ ktab, int Kpad, int m0,
                                          const LaneCtx& c, float* as /* [2][16][BM] */,
                                          int4* kts /* [3][16] */, f32x16 (&acc)[R]) {
  const int tid = threadIdx.x;
  const int h = (tid & 63) >> 5;
  const int nk = Kpad / 16;
  if (nk == 0) return;
  const int4* kt4 = reinterpret_cast<const int4*>(ktab);
  float b0[8], b1[8];
  f32x4 areg[2];
  stage_a_load<R>(apack, d.Mpad, 0, m0, tid, areg);
  if (tid < 32) kts[tid] = kt4[tid];     // table rows of K steps 0 and 1 (table is padded)
  stage_a_store<R>(as, tid, areg);
  __syncthreads();
  load_b8(d, kts, h, c, b0);
  int ks = 0;
  for (; ks + 1 < nk; ks += 2) {
    k_step<R>(d, apack, kt4, ks, m0, c, as, kts, acc, b0, b1);
    k_step<R>(d, apack, kt4, ks + 1, m0, c, as, kts, acc, b1, b0);
  }
  if (ks < nk) k_step<R>(d, apack, kt4, ks, m0, c, as, kts, acc, b0, b1);
}

// Position / tile bookkeeping shared by the forward kernels and their epilogue.
struct TileCtx {
  int n, pw, m0, a, b, wave, lane, l31, h;
  bool jvalid;
};

// bias + activation applied to the accumulators (between the phases of a two-phase launch).  All per-row loads are
// issued TOGETHER, unconditionally (clamped row index) and under wave-uniform tests only: with `if (e.bias) v += e.bias[m]`
// inside the per-element loop hipcc emitted one load + s_waitcnt vmcnt(0) per element, i.e. 16*R serial L2 round trips
// (~40 us) per workgroup -- more than the whole K loop of the short-K layers.
// FULL = false (persistent short-K kernel of gemm_tap.h): bias, residual, fused GLU store and statistics only -- the
// launcher routes activations, the backward-of-activation mode and phase-merged stores to the tiled kernels, so those
// paths (and their registers) are compiled out of the tile loop.
// the per-row bias values of a lane (clamped row index; merged phases: bias per channel; fused GLU: rows interleaved
// (2c, 2c+1) <-> channels (c, C+c))
template <int R>
__device__ __forceinline__ void fwd_load_bias(const FwdArgs& g, int m0, int h, float (&bv)[R][16]) {
  const rfx_gemm_desc& d = g.d;
  const rfx_epilogue& e = g.e;
#pragma unroll
  for (int mt = 0; mt < R; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const int mc = m < d.M ? m : d.M - 1;
      bv[mt][r] = e.bias[e.glu_out ? (mc & 1) * (d.M >> 1) + (mc >> 1) : (mc >> d.mg_log)];
    }
}

template <int R, bool FULL = true>
__device__ __forceinline__ void fwd_epilogue_mid(const FwdArgs& g, const TileCtx& tc, f32x16 (&acc)[R],
                                                 const float (*pre)[16] = nullptr) {
  const rfx_gemm_desc& d = g.d;
  const rfx_epilogue& e = g.e;
  const int m0 = tc.m0, h = tc.h;
  {
    float bv[R][16];
    if (e.bias) {
      if (pre) {                       // values loaded once by the caller (persistent kernel: no load inside the tile loop)
#pragma unroll
        for (int mt = 0; mt < R; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) bv[mt][r] = pre[mt][r];
      } else {
        fwd_load_bias<R>(g, m0, h, bv);
      }
#pragma unroll
      for (int mt = 0; mt < R; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] += bv[mt][r];
    }
    if (FULL && e.act != RFX_ACT_NONE && !e.bwd) {
      if (e.act == RFX_ACT_PRELU) {
#pragma unroll
        for (int mt = 0; mt < R; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            bv[mt][r] = e.act_param[m < d.M ? m : d.M - 1];
          }
#pragma unroll
        for (int mt = 0; mt < R; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][r] = acc[mt][r] >= 0.f ? acc[mt][r] : bv[mt][r] * acc[mt][r];
      } else {
#pragma unroll
        for (int mt = 0; mt < R; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][r] = rfx_act_apply(acc[mt][r], e.act, 0.f);
      }
    }
  }
}

// everything after the last K loop: residual / second activation / the store variants / statistics
// 16-bit stores two positions at a time: lanes (j, j + 1), j even, swap one value (DPP quad_perm [1, 0, 3, 2]) so that the even lane
// holds row r0 of both positions and the odd lane row r1 of both -- one dword store per lane and row pair instead of two 2-byte
// stores (a 2-byte store per lane makes every store instruction carry 64 B; the bf16-output launches were store-issue-bound).
__device__ __forceinline__ void bf16_pair_store(uint16_t* own_pos, bool odd, int64_t row0_off, int64_t row1_off, bool ok0, bool ok1,
                                                uint32_t b0, uint32_t b1) {
  const uint32_t send = odd ? b0 : b1;
  const uint32_t recv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)send, 0xB1, 0xF, 0xF, false);
  const uint32_t word = odd ? (recv | (b1 << 16)) : (b0 | (recv << 16));
  uint16_t* p = odd ? own_pos - 1 + row1_off : own_pos + row0_off;
  if (odd ? ok1 : ok0) *reinterpret_cast<uint32_t*>(p) = word;
}

// ---- lean stores of a FULL tile ------------------------------------------------------------------------------------
// r03 PMC on the streaming kernel (12 -> 96 channels, 1x1, bf16 output): 392 VALU instructions per 32x32 tile, SQ_ACTIVE_INST_VALU =
// 57 % of the kernel's time, 2.3 TB/s -- the "bandwidth-class" launches were VALU-bound in the generic epilogue below (per-row 64-bit
// address arithmetic, per-row validity branches, a 4-instruction software bf16 rounding per value).  When every position of the wave's
// tile is valid and all 32 R rows are < M (a wave-uniform test) the per-row part of each address is a SCALAR (the soffset operand of a
// raw buffer store: (row constant) * channel stride), the per-lane part is computed once per tile, and the rounding is
// v_cvt_pk_bf16_f32 (round-to-nearest-even, bit-equal to bf16_rne on finite values).
// Launch-uniform part of the test (also evaluated on the host: the streaming kernel only takes launches whose EVERY tile qualifies).
// *records = the buffer's num_records: rows >= M of a partial channel tile are dropped by the hardware range check when every position
// offset is < the channel stride (a plain NCHW tensor) -- offset = m * cs + position >= M * cs  <=>  m >= M
static inline __host__ __device__ bool rfx_fast_store_geo(const rfx_gemm_desc& d, int esz, uint32_t* records, bool* partial_ok) {
  const int64_t posmax = (int64_t)((d.OA - 1) * d.out_sa + d.out_a0) * d.out_as + (int64_t)((d.OB - 1) * d.out_sb + d.out_b0) * d.out_bs;
  const int64_t maxoff = (int64_t)(d.M - 1) * d.out_cs + posmax;
  *partial_ok = posmax < d.out_cs && !(d.M & 1);
  *records = *partial_ok ? (uint32_t)((int64_t)d.M * d.out_cs * esz) : 0x7fffffffu;
  return d.out_cs > 0 && d.out_as >= 0 && d.out_bs >= 0 && (maxoff + 2) * esz < 0x7fffffffll;
}
// 16-bit output: positions j, j + 1 (j even) are adjacent elements of one row, 4-byte aligned, and valid together
static inline __host__ __device__ bool rfx_pair16_geo(const rfx_gemm_desc& d, const void* out) {
  return d.out_bf16 && d.out_bs == 1 && d.out_sb == 1 && !((d.OB | d.out_b0) & 1) && !((d.out_ns | d.out_cs | d.out_as) & 1) &&
         ((reinterpret_cast<uintptr_t>(out) & 3) == 0);
}
// does the wave's whole tile qualify?  (wave-uniform)
__device__ __forceinline__ bool fwd_tile_full(const rfx_gemm_desc& d, const TileCtx& tc, int rows, int esz, uint32_t* records = nullptr) {
  uint32_t rec; bool partial_ok;
  const bool geo = rfx_fast_store_geo(d, esz, &rec, &partial_ok);
  const bool rows_ok = tc.m0 + rows <= d.M || (records && partial_ok);
  if (records) *records = rec;
  return geo && rows_ok && __builtin_amdgcn_ballot_w64(!tc.jvalid) == 0;
}

// plain store of a full tile (+ the moments s1, s2 of the stored values).  pair16: 16-bit output, two positions per store -- even
// lanes write row r of positions (j, j + 1), odd lanes row r + 1 of positions (j - 1, j); otherwise fp32.  records: see fwd_tile_full.
template <int R>
__device__ __forceinline__ void fwd_store_fast_plain(const FwdArgs& g, const TileCtx& tc, const f32x16 (&acc)[R], int64_t opos, bool pair16,
                                                     uint32_t records, float& s1, float& s2) {
  const rfx_gemm_desc& d = g.d;
  const int m0 = tc.m0, h = tc.h;
  const bool partial = m0 + 32 * R > d.M;               // wave-uniform: rows >= M are dropped by the range check, masked out of the sums
  char* base = reinterpret_cast<char*>(g.out) + (int64_t)tc.n * d.out_ns * (d.out_bf16 ? 2 : 4);
  const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)records, 0x00020000);
  if (pair16) {
    const bool odd = tc.l31 & 1;
    const uint32_t csb = (uint32_t)d.out_cs * 2u;
    const uint32_t vo = (uint32_t)(opos * 2) + (uint32_t)(4 * h) * csb + (odd ? csb - 2u : 0u);
    const uint32_t sel = odd ? 0x03020706u : 0x05040100u;
#pragma unroll
    for (int mt = 0; mt < R; ++mt)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const uint32_t mc = (uint32_t)__builtin_amdgcn_readfirstlane(m0 + mt * 32 + (r & 3) + 8 * (r >> 2));
        const uint32_t w = rfx_cvt_pk_bf16(acc[mt][r], acc[mt][r + 1]);
        const uint32_t recv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0xB1, 0xF, 0xF, false);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_amdgcn_perm(recv, w, sel), rso, vo, mc * csb, 0);
        float v0 = __uint_as_float(w << 16), v1 = __uint_as_float(w & 0xffff0000u);
        if (partial && (int)mc + 4 * h >= d.M) { v0 = 0.f; v1 = 0.f; }
        s1 += v0 + v1; s2 += v0 * v0 + v1 * v1;
      }
  } else {
    const uint32_t csb = (uint32_t)d.out_cs * 4u;
    const uint32_t vo = (uint32_t)(opos * 4) + (uint32_t)(4 * h) * csb;
#pragma unroll
    for (int mt = 0; mt < R; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t mc = (uint32_t)__builtin_amdgcn_readfirstlane(m0 + mt * 32 + (r & 3) + 8 * (r >> 2));
        float v = acc[mt][r];
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rso, vo, mc * csb, 0);
        if (partial && (int)mc + 4 * h >= d.M) v = 0.f;
        s1 += v; s2 += v * v;
      }
  }
}
// fused-GLU store of a full tile with a 16-bit conv output: rows (r, r + 1) of a lane are (a, b) of GLU channel ch = m >> 1.  Even lanes
// store the a row (channel ch) of positions (j, j + 1), odd lanes the b row (channel Ch + ch) of positions (j - 1, j); every lane stores
// a * sigmoid(b) of its own position (fp32), taken of the STORED (rounded) values as the backward pass will see them.
template <int R>
__device__ __forceinline__ void fwd_store_fast_glu(const FwdArgs& g, const TileCtx& tc, const f32x16 (&acc)[R], int64_t opos) {
  const rfx_gemm_desc& d = g.d;
  const int m0 = tc.m0, h = tc.h, Ch = d.M >> 1;
  const bool odd = tc.l31 & 1;
  const uint32_t csb = (uint32_t)d.out_cs * 2u;
  const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(g.out) + (int64_t)tc.n * d.out_ns * 2, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(g.e.glu_out) + (int64_t)tc.n * g.e.glu_ns * 4, 0, 0x7fffffff, 0x00020000);
  const uint32_t vo = (uint32_t)(opos * 2) + (uint32_t)(2 * h) * csb + (odd ? (uint32_t)Ch * csb - 2u : 0u);
  const uint32_t vg = (uint32_t)(opos * 4) + (uint32_t)(2 * h) * csb * 2u;
  const uint32_t sel = odd ? 0x03020706u : 0x05040100u;
#pragma unroll
  for (int mt = 0; mt < R; ++mt)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const uint32_t chc = (uint32_t)__builtin_amdgcn_readfirstlane((m0 + mt * 32) / 2 + ((r & 3) >> 1) + 4 * (r >> 2));
      const uint32_t w = rfx_cvt_pk_bf16(acc[mt][r], acc[mt][r + 1]);
      const uint32_t recv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0xB1, 0xF, 0xF, false);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_amdgcn_perm(recv, w, sel), rso, vo, chc * csb, 0);
      const float av = __uint_as_float(w << 16), bv = __uint_as_float(w & 0xffff0000u);
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(av * rfx_sigmoid(bv)), rsg, vg, chc * csb * 2u, 0);
    }
}

template <int R, bool FULL = true>
__device__ __forceinline__ void fwd_epilogue_store(const FwdArgs& g, const TileCtx& tc, f32x16 (&acc)[R]) {
  const rfx_gemm_desc& d = g.d;
  const rfx_epilogue& e = g.e;
  const int n = tc.n, pw = tc.pw, m0 = tc.m0, a = tc.a, b = tc.b, wave = tc.wave, lane = tc.lane, l31 = tc.l31, h = tc.h;
  struct { bool jvalid; } c = {tc.jvalid};
  const int64_t opos = (int64_t)(a * d.out_sa + d.out_a0) * d.out_as + (int64_t)(b * d.out_sb + d.out_b0) * d.out_bs;
  float* outp = g.out + (int64_t)n * d.out_ns + opos;
  uint16_t* outh = reinterpret_cast<uint16_t*>(g.out) + (int64_t)n * d.out_ns + opos;      // d.out_bf16: same element offsets
  // positions j, j + 1 (j even) are adjacent 16-bit elements of one row and valid together (wave-uniform test)
  const bool pair16 = g.pair_store && rfx_pair16_geo(d, g.out);
  const bool odd = l31 & 1;
  const float* resp = nullptr;
  if (e.res)
    resp = e.res + (int64_t)n * e.res_ns + (int64_t)(a * d.out_sa + d.out_a0) * e.res_as +
           (int64_t)(b * d.out_sb + d.out_b0) * e.res_bs;
  if (FULL && e.bwd) {
    // out = G * act'(pre);  gparam[m] += sum_j G * min(pre, 0)   (PReLU slope gradient).  Loads batched as above.
    float gin[R][16];                            // all incoming gradients in flight at once; slopes a channel tile at a time
#pragma unroll
    for (int mt = 0; mt < R; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        gin[mt][r] = resp[(int64_t)(m < d.M ? m : d.M - 1) * e.res_cs];
      }
#pragma unroll
    for (int mt = 0; mt < R; ++mt) {
      float sl[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        sl[r] = (e.act == RFX_ACT_PRELU) ? e.act_param[m < d.M ? m : d.M - 1] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        float gs = 0.f;
        if (m < d.M && c.jvalid) {
          const float pre = acc[mt][r];
          outp[(int64_t)m * d.out_cs] = gin[mt][r] * rfx_act_grad(pre, e.act, sl[r]);
          gs = pre < 0.f ? gin[mt][r] * pre : 0.f;
        }
        if (e.gparam) {   // reduce over the 32 position lanes of this half-wave
#pragma unroll
          for (int o = 16; o > 0; o >>= 1) gs += __shfl_xor(gs, o, 64);
          // stat_slots > 1: gparam is [slots][M] partial sums (the caller adds them up): every wave of every
          // workgroup adds into the same M addresses otherwise (2.6e5-way contention per address on the TCN shapes)
          if (l31 == 0 && m < d.M)
            atomicAdd(e.gparam + (int64_t)(e.stat_slots > 1 ? (pw * 4 + wave) & (e.stat_slots - 1) : 0) * d.M + m, gs);
        }
      }
    }
    return;
  }
  if (FULL && d.mg_log) {
    // phase-merged store (see rfx_gemm_desc.mg_*): row m = channel*G + phase, position index i on the merged axis ->
    // axis index i*G + phase + mg_off.  A lane's 4 consecutive rows (r & 3) are the 4 phases of one channel when G = 4:
    // consecutive output samples, and the 32 lanes cover 32 consecutive position indices -> full lines per wave.
    const int G1 = (1 << d.mg_log) - 1;
    const int pos = d.mg_axis ? b : a;
    const int64_t other = d.mg_axis ? (int64_t)(a * d.out_sa + d.out_a0) * d.out_as : (int64_t)(b * d.out_sb + d.out_b0) * d.out_bs;
    const int64_t st = d.mg_axis ? d.out_bs : d.out_as;
    float* ob = g.out + (int64_t)n * d.out_ns + other;
    // optional residual (a second gradient path into the same tensor: HDemucs skip connections), same coordinates as out
    const int64_t rother = d.mg_axis ? (int64_t)(a * d.out_sa + d.out_a0) * e.res_as : (int64_t)(b * d.out_sb + d.out_b0) * e.res_bs;
    const int64_t rst = d.mg_axis ? e.res_bs : e.res_as;
    const float* rb = e.res ? e.res + (int64_t)n * e.res_ns + rother : nullptr;
    if (d.mg_log == 2 && st == 1 && (!rb || rst == 1)) {
      // G = 4, unit stride along the merged axis: a lane's rows r..r+3 are the 4 phases of one channel = 4 CONSECUTIVE output
      // samples -> one 16-byte store (and one 16-byte residual load) per lane and channel when all four lie inside the row
#pragma unroll
      for (int mt = 0; mt < R; ++mt)
#pragma unroll
        for (int r = 0; r < 16; r += 4) {
          const int m = m0 + mt * 32 + 8 * (r >> 2) + 4 * h;          // phase 0 row of this quad (m0, 32 mt, 8 (r>>2), 4 h: all multiples of 4)
          const int idx = (pos << 2) + d.mg_off;
          if (!c.jvalid || m >= d.M) continue;
          float* op = ob + (int64_t)(m >> 2) * d.out_cs + idx;
          const float* rp = rb ? rb + (int64_t)(m >> 2) * e.res_cs + idx : nullptr;
          if (idx >= 0 && idx + 3 < d.mg_len) {
            f32x4 v = {acc[mt][r], acc[mt][r + 1], acc[mt][r + 2], acc[mt][r + 3]};
            if (rp) {
              f32x4 rv;
              __builtin_memcpy(&rv, rp, 16);                         // 4-byte aligned 16-byte access (mg_off need not be a multiple of 4)
              v += rv;
            }
            __builtin_memcpy(op, &v, 16);
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if ((unsigned)(idx + q) < (unsigned)d.mg_len) op[q] = acc[mt][r + q] + (rp ? rp[q] : 0.f);
          }
        }
      return;
    }
    if (rb) {          // wave-uniform.  All residual loads of a channel tile first (clamped addresses, unconditional), then the adds:
                       // interleaved with the stores they would be serialised by the may-alias ordering
#pragma unroll
      for (int mt = 0; mt < R; ++mt) {
        float rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const int idx = (pos << d.mg_log) + (m & G1) + d.mg_off;
          const bool ok = c.jvalid && m < d.M && (unsigned)idx < (unsigned)d.mg_len;
          rv[r] = rb[ok ? (int64_t)(m >> d.mg_log) * e.res_cs + (int64_t)idx * rst : 0];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] += rv[r];
      }
    }
#pragma unroll
    for (int mt = 0; mt < R; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int idx = (pos << d.mg_log) + (m & G1) + d.mg_off;
        if (c.jvalid && m < d.M && (unsigned)idx < (unsigned)d.mg_len)
          ob[(int64_t)(m >> d.mg_log) * d.out_cs + (int64_t)idx * st] = acc[mt][r];
      }
    return;
  }
  if (e.glu_out) {
    // rows (r, r+1) of a lane are (a, b) of one GLU channel c = m >> 1: conv output in natural order + a * sigmoid(b)
    const int Ch = d.M >> 1;
    float* gl = e.glu_out + (int64_t)n * e.glu_ns + opos;
    if (pair16 && fwd_tile_full(d, tc, 32 * R, 4)) {
      fwd_store_fast_glu<R>(g, tc, acc, opos);
      return;
    }
    if (pair16) {                     // all lanes take part in the exchange; validity only gates the stores
#pragma unroll
      for (int mt = 0; mt < R; ++mt)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const int ch = m >> 1;
          const bool ok = c.jvalid && m < d.M;
          const uint32_t ab = bf16_rne(acc[mt][r]), bb = bf16_rne(acc[mt][r + 1]);
          bf16_pair_store(outh, odd, (int64_t)ch * d.out_cs, (int64_t)(Ch + ch) * d.out_cs, ok, ok, ab, bb);
          if (ok) gl[(int64_t)ch * d.out_cs] = __uint_as_float(ab << 16) * rfx_sigmoid(__uint_as_float(bb << 16));
        }
      return;
    }
    if (c.jvalid) {
#pragma unroll
      for (int mt = 0; mt < R; ++mt)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (m < d.M) {
            const int ch = m >> 1;
            float a = acc[mt][r], b = acc[mt][r + 1];
            if (d.out_bf16) {      // conv output kept in 16 bits (wave-uniform); the GLU is taken of the STORED values, as the
              const uint32_t ab = bf16_rne(a), bb = bf16_rne(b);     // backward pass will see them
              outh[(int64_t)ch * d.out_cs] = (uint16_t)ab;
              outh[(int64_t)(Ch + ch) * d.out_cs] = (uint16_t)bb;
              a = __uint_as_float(ab << 16); b = __uint_as_float(bb << 16);
            } else {
              outp[(int64_t)ch * d.out_cs] = a;
              outp[(int64_t)(Ch + ch) * d.out_cs] = b;
            }
            gl[(int64_t)ch * d.out_cs] = a * rfx_sigmoid(b);
          }
        }
    }
    return;
  }
  float s1 = 0.f, s2 = 0.f;      // optional per-sample moments of the stored values (GroupNorm(1, C) statistics)
  if (resp) {                    // wave-uniform; residual values fetched a channel tile at a time (see the bias note)
#pragma unroll
    for (int mt = 0; mt < R; ++mt) {
      float rv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        rv[r] = resp[(int64_t)(m < d.M ? m : d.M - 1) * e.res_cs];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][r] += rv[r];
    }
  }
  if (FULL && e.act2 != RFX_ACT_NONE) {
#pragma unroll
    for (int mt = 0; mt < R; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][r] = rfx_act_apply(acc[mt][r], e.act2, 0.f);
  }
  uint32_t frec = 0;
  if ((pair16 || !d.out_bf16) && fwd_tile_full(d, tc, 32 * R, d.out_bf16 ? 2 : 4, &frec)) {
    fwd_store_fast_plain<R>(g, tc, acc, opos, pair16, frec, s1, s2);
  } else
  if (pair16) {
#pragma unroll
    for (int mt = 0; mt < R; ++mt)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;         // rows m, m + 1
        const bool ok0 = c.jvalid && m < d.M, ok1 = c.jvalid && m + 1 < d.M;
        const uint32_t b0 = bf16_rne(acc[mt][r]), b1 = bf16_rne(acc[mt][r + 1]);
        bf16_pair_store(outh, odd, (int64_t)m * d.out_cs, (int64_t)(m + 1) * d.out_cs, ok0, ok1, b0, b1);
        const float v0 = ok0 ? __uint_as_float(b0 << 16) : 0.f, v1 = ok1 ? __uint_as_float(b1 << 16) : 0.f;
        s1 += v0 + v1; s2 += v0 * v0 + v1 * v1;
      }
  } else if (c.jvalid) {
#pragma unroll
    for (int mt = 0; mt < R; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < d.M) {
          float v = acc[mt][r];
          if (d.out_bf16) {        // wave-uniform; statistics are those of the stored (rounded) values
            const uint32_t vb = bf16_rne(v);
            outh[(int64_t)m * d.out_cs] = (uint16_t)vb;
            v = __uint_as_float(vb << 16);
          } else {
            outp[(int64_t)m * d.out_cs] = v;
          }
          s1 += v; s2 += v * v;
        }
      }
    }
  }
  if (e.stat_sums) {               // wave-uniform branch: one fp64 atomic pair per wave
    const double d1 = rfx_wave_sum_d((double)s1), d2 = rfx_wave_sum_d((double)s2);
    const int slots = e.stat_slots > 1 ? e.stat_slots : 1;
    double* dst = e.stat_sums + 2 * ((int64_t)n * slots + (pw & (slots - 1)));
    if (lane == 0) { atomicAdd(dst, d1); atomicAdd(dst + 1, d2); }
  }
}

template <int R>
__global__ __launch_bounds__(256, 2) void gemm_fwd_kernel(const FwdArgs g) {
  constexpr int BM = 32 * R;
  __shared__ __attribute__((aligned(16))) float as[2 * 16 * BM];
  __shared__ __attribute__((aligned(16))) int4 kts[3 * 16];
  const rfx_gemm_desc& d = g.d;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int P = d.OA * d.OB;
  // XCD-aware tile order (block b runs on XCD b % 8, each XCD has its own L2): the channel tiles of one
  // (sample, position tile) read the same input samples, so they are made consecutive ON THE SAME XCD;
  // neighbouring position tiles are spread over the 8 XCDs.
  const int mtiles = d.Mpad / BM, ptiles = (P + 127) / 128;
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int ym = q % mtiles;
  const int pw = g.xcd_chunk > 0 ? xcd * g.xcd_chunk + q / mtiles : (q / mtiles) * 8 + xcd;   // (n, position tile) work item
  if (pw >= ptiles * d.N) return;
  const int n = pw / ptiles;
  const int m0 = ym * BM;
  const int j = (pw - n * ptiles) * 128 + wave * 32 + l31;
  LaneCtx c;
  c.jvalid = j < P;
  const int jj = c.jvalid ? j : 0;
  const int a = jj / d.OB, b = jj - a * d.OB;
  c.ia0 = a * d.SA;
  c.ib0 = b * d.SB;
  c.safe = rfx_zero_f32;
  c.inb = g.in + (int64_t)n * d.in_ns + (int64_t)c.ia0 * d.in_as + (int64_t)c.ib0 * d.in_bs;
  c.rs = rfx_sample_rsrc(g.in + (int64_t)n * d.in_ns);
  c.voff = (uint32_t)(((int64_t)c.ia0 * d.in_as + (int64_t)c.ib0 * d.in_bs) * 4);

  f32x16 acc[R];
#pragma unroll
  for (int mt = 0; mt < R; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  run_phase<R>(d, g.apack, g.ktab, d.Kpad, m0, c, as, kts, acc);

  const bool two = g.apack2 != nullptr;
  TileCtx tc;
  tc.n = n; tc.pw = pw; tc.m0 = m0; tc.a = a; tc.b = b; tc.wave = wave; tc.lane = lane; tc.l31 = l31; tc.h = h;
  tc.jvalid = c.jvalid;
  fwd_epilogue_mid<R>(g, tc, acc);
  if (two) {
    LaneCtx c2 = c;
    if (g.in2) {
      c2.inb = g.in2 + (c.inb - g.in);
      c2.rs = rfx_sample_rsrc(g.in2 + (int64_t)n * d.in_ns);
    }
    __syncthreads();            // phase 1 may still be reading the LDS buffers
    run_phase<R>(d, g.apack2, g.ktab2, g.Kpad2, m0, c2, as, kts, acc);
  }

  fwd_epilogue_store<R>(g, tc, acc);
}

// launchers of the tiled forward kernels: channel-major table, exact fp32 (gemm_fwd_f32.hip); tap-major, split bf16x3
// (gemm_fwd_bf3.hip) and plain bf16 operands (gemm_fwd_bf16.hip) -- see gemm_tap.h
int rfx_launch_gemm_fwd_f32(const FwdArgs& g, int r, dim3 grid, hipStream_t s);
// short single-phase reductions with enough position tiles to keep persistent workgroups busy run the streaming kernel
// (gemm_tap_stream_kernel); shared by the launcher and rfx_gemm_fwd_variant
// ... and whose every wave tile takes the lean store (stream_store): 32 | positions per sample, rows full or dropped by the range check,
// 16-bit outputs pairable.  (The output pointer's 4-byte alignment is checked by the launcher; every torch allocation has it.)
static inline bool rfx_tap_use_stream(const rfx_gemm_desc& d, const rfx_epilogue& e, bool two_phase, int r) {
  const int64_t work = (int64_t)((d.OA * d.OB + 127) / 128) * d.N;
  uint32_t rec; bool partial_ok;
  const bool geo = rfx_fast_store_geo(d, (d.out_bf16 && !e.glu_out) ? 2 : 4, &rec, &partial_ok);
  const bool rows_ok = d.M % 32 == 0 || (partial_ok && !e.glu_out);
  const bool store_ok = geo && rows_ok && (d.OA * d.OB) % 32 == 0 && (d.out_bf16 ? rfx_pair16_geo(d, nullptr) : !e.glu_out);
  return d.in_bf16 == 0 && d.Kpad_t <= 64 && !two_phase && work >= 4096 && r == 1 && e.act == RFX_ACT_NONE &&
         e.act2 == RFX_ACT_NONE && !e.bwd && d.mg_log == 0 && !e.res && store_ok;
}
int rfx_launch_gemm_fwd_bf3(const FwdArgs& g, int r, dim3 grid, hipStream_t s);
// halo-tile kernel (gemm_halo.h, gemm_fwd_halo.hip): launch-uniform eligibility, shared by the launcher and rfx_gemm_fwd_variant
#define RFX_HALO_TW 128
static inline __host__ __device__ bool rfx_halo_geo_ok(const rfx_gemm_desc& d) {
  if (d.halo_nt != 3 && d.halo_nt != 9) return false;
  const int slots = d.halo_nt == 9 ? 2 : 1;
  return d.SA == 1 && d.SB == 1 && d.in_bs == 1 && d.mg_log == 0 && d.OB % RFX_HALO_TW == 0 && d.halo_rows >= 1 &&
         d.halo_rows <= 3 && d.halo_w >= RFX_HALO_TW && d.halo_rows * d.halo_w <= 256 * slots && d.gpt >= 2 && (d.gpt & 1) == 0 &&
         d.ntaps % d.halo_nt == 0 && d.Kpad_t == 8 * d.ntaps * d.gpt;
}
static inline int rfx_halo_pick_r(const rfx_gemm_desc& d) {
  if (d.Mpad % 96 == 0) return 3;
  if (d.Mpad % 64 == 0) return 2;
  return d.Mpad % 32 == 0 ? 1 : 0;
}
static inline bool rfx_halo_takes(const rfx_gemm_desc& d) { return d.R > 0 && rfx_halo_geo_ok(d) && rfx_halo_pick_r(d) > 0; }
int rfx_launch_gemm_halo(const FwdArgs& g, hipStream_t s);
int rfx_launch_gemm_fwd_bf16(const FwdArgs& g, int r, dim3 grid, hipStream_t s);
