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
};


typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t bf16_rne(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

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

// ---------------------------------------------------------------------------------
#define RFX_BDIST 3   // gather look-ahead in K steps (ktab ring: 8 slots; tables are padded by 96 rows)
// bf16x3 variant of the K loop: every fp32 operand is split x = hi + lo (two bf16) and
// a.b ~= hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_bf16 (fp32 accumulate): ~2^-16
// relative error per product instead of 2^-24, at 3/16 of the fp32-MFMA issue cost.
// Lane (j = lane & 31, h = lane >> 5) now gathers the 8 consecutive taps k = 8h .. 8h+7 of its
// column (one MFMA B fragment); packed weights arrive pre-split (pack_a_bf3_kernel).
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void load_b8_bf3(const rfx_gemm_desc& d, const int4* ktl, int h, const LaneCtx& c,
                                            float (&b)[8]) {
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    int4 e[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) e[q] = ktl[8 * h + half * 4 + q];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool ok = c.jvalid & ((unsigned)(c.ia0 + e[q].y) < (unsigned)d.IA) &
                      ((unsigned)(c.ib0 + e[q].z) < (unsigned)d.IB);
      const uint32_t off = ok ? c.voff + ((uint32_t)e[q].x << 2) : RFX_BUF_OOB;
      b[half * 4 + q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c.rs, off, 0, 0));
    }
  }
}

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// x = hi + lo with hi = RNE_bf16(x), lo = RNE_bf16(x - hi): v_cvt_pk_bf16_f32 does two values per instruction and
// the residual is one packed subtract -> 5 VALU per pair (the mask / shift / add sequence it replaces took 13)
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
  uint32_t hw[4], lw[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x2_t v = {x[2 * q], x[2 * q + 1]};
    const uint32_t h = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
    const f32x2_t hf = {__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
    hw[q] = h;
    lw[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v - hf, bf16x2_t));
  }
  hi = __builtin_bit_cast(bf16x8, make_uint4(hw[0], hw[1], hw[2], hw[3]));
  lo = __builtin_bit_cast(bf16x8, make_uint4(lw[0], lw[1], lw[2], lw[3]));
}

// A stage of the bf16x3 path: (hi, lo) x 2 k8 rows x BM cells = 4*BM 16-byte cells per K step, one or two per thread.
// Plain scalars (not arrays) so the two in-flight stages of the software pipeline stay in registers.
struct AStage { uint4 v0, v1; };
template <int R>
__device__ __forceinline__ uint4 stage_a_bf3_cell(const uint4* __restrict__ apk, int64_t arr_stride, int Mpad,
                                                  int k8_0, int m0, int idx) {
  constexpr int BM = 32 * R, NV = 4 * BM;
  idx = idx < NV ? idx : NV - 1;
  const int arr = idx / (2 * BM), rem = idx % (2 * BM);
  const int kk8 = rem / BM, mm = rem % BM;
  return apk[arr * arr_stride + (int64_t)(k8_0 + kk8) * Mpad + m0 + mm];
}
template <int R>
__device__ __forceinline__ AStage stage_a_bf3_load(const uint4* __restrict__ apk, int64_t arr_stride, int Mpad,
                                                   int k8_0, int m0, int tid) {
  constexpr int NV = 128 * R;
  AStage s;
  s.v0 = stage_a_bf3_cell<R>(apk, arr_stride, Mpad, k8_0, m0, tid);
  s.v1 = NV > 256 ? stage_a_bf3_cell<R>(apk, arr_stride, Mpad, k8_0, m0, tid + 256) : s.v0;
  return s;
}
template <int R>
__device__ __forceinline__ void stage_a_bf3_store(uint4* as, int tid, const AStage& s) {
  constexpr int NV = 128 * R;
  // UNCONDITIONAL stores (surplus threads rewrite the last cell with the same data), see stage_a_store
  as[tid < NV ? tid : NV - 1] = s.v0;
  if (NV > 256) as[tid + 256 < NV ? tid + 256 : NV - 1] = s.v1;
}

// One 16-deep K step of the bf16x3 pipeline.  LDS holds the A tiles of TWO K steps per buffer, so the workgroup
// barrier comes only after every odd step (SUB == 1): with one barrier per K step the four waves re-synchronised every
// ~0.2 us of matrix work and 40 % of the wave time was parked (PMC); a race-y run with half the barriers
// bounded the gain at 3 % of the Demucs step.  Per step ks:
//   LDS -> fragments of A(ks) from buffer (ks/2)&1, half SUB;
//   global -> registers: table row of step ks+5, A tile of step ks+4 (into the register set that held A(ks+2));
//   gathers of step ks+3 (the 4-deep ring);  MFMAs;
//   registers -> LDS: A(ks+2) into the OTHER buffer, table row into ring slot (ks+5)&7;  barrier if SUB.
// Everything written in steps {2D, 2D+1} is first read in step 2D+2, i.e. behind the barrier that ends step 2D+1.
template <int R, int SUB>
__device__ __forceinline__ void k_step_bf3(const rfx_gemm_desc& d, const uint4* __restrict__ apk,
                                           int64_t arr_stride, const int4* __restrict__ kt4, int ks, int m0,
                                           const LaneCtx& c, uint4* as, int4* kts, f32x16 (&acc)[R],
                                           const float (&bc)[8], float (&bn)[8], AStage& a_set) {
  constexpr int BM = 32 * R;
  const int tid = threadIdx.x;
  const int lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int buf = (ks >> 1) & 1;
  const uint4* a_lds = as + buf * 8 * BM + SUB * 4 * BM;
  uint4 ah[R], al[R];
#pragma unroll
  for (int mt = 0; mt < R; ++mt) {
    ah[mt] = a_lds[h * BM + mt * 32 + l31];
    al[mt] = a_lds[2 * BM + h * BM + mt * 32 + l31];
  }
  // issue order matters: vmcnt retires in order; the table row is written to LDS at the end of THIS step, so it goes
  // first; the gathers, consumed RFX_BDIST steps later, go last and stay in flight
  const int4 ktreg = kt4[(ks + RFX_BDIST + 2) * 16 + (tid & 15)];
  const AStage a_now = a_set;                                         // A(ks+2), fetched two steps ago
  a_set = stage_a_bf3_load<R>(apk, arr_stride, d.Mpad, 2 * (ks + 4), m0, tid);
  load_b8_bf3(d, kts + ((ks + RFX_BDIST) & 7) * 16, h, c, bn);
  bf16x8 bh, bl;
  split8(bc, bh, bl);
#pragma unroll
  for (int mt = 0; mt < R; ++mt) {
    const bf16x8 fh = __builtin_bit_cast(bf16x8, ah[mt]), fl = __builtin_bit_cast(bf16x8, al[mt]);
    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, bh, acc[mt], 0, 0, 0);
    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, bl, acc[mt], 0, 0, 0);
    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl, bh, acc[mt], 0, 0, 0);
  }
  stage_a_bf3_store<R>(as + (buf ^ 1) * 8 * BM + SUB * 4 * BM, tid, a_now);
  kts[((ks + RFX_BDIST + 2) & 7) * 16 + (tid & 15)] = ktreg;   // every thread (same value per tid & 15)
  if (SUB) __syncthreads();
  else { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }   // keep the two steps apart in the compiler too
}

template <int R>
__device__ __forceinline__ void run_phase_bf3(const rfx_gemm_desc& d, const float* __restrict__ apack,
                                              const rfx_ktab_entry* __restrict__ ktab, int Kpad, int m0,
                                              const LaneCtx& c, float* as_f, int4* kts, f32x16 (&acc)[R]) {
  constexpr int BM = 32 * R;
  const int tid = threadIdx.x;
  const int h = (tid & 63) >> 5;
  const int nk = Kpad / 16;
  if (nk == 0) return;
  const uint4* apk = reinterpret_cast<const uint4*>(apack);
  const int64_t arr_stride = (int64_t)(Kpad / 8 + 8) * d.Mpad;
  uint4* as = reinterpret_cast<uint4*>(as_f);
  const int4* kt4 = reinterpret_cast<const int4*>(ktab);
  __syncthreads();            // a previous phase (two-phase launches) may still be reading the LDS buffers
  float b0[8], b1[8];
  {
    const AStage s0 = stage_a_bf3_load<R>(apk, arr_stride, d.Mpad, 0, m0, tid);
    const AStage s1 = stage_a_bf3_load<R>(apk, arr_stride, d.Mpad, 2, m0, tid);
    if (tid < 16 * (RFX_BDIST + 2)) kts[tid] = kt4[tid];   // table rows of the first K steps (table is padded)
    stage_a_bf3_store<R>(as, tid, s0);                       // A(0), A(1) -> buffer 0
    stage_a_bf3_store<R>(as + 4 * BM, tid, s1);
  }
  __syncthreads();
  // the gathers run RFX_BDIST K steps ahead of the MFMAs: one K step is ~0.2 us of matrix work, a gather that
  // misses L2 takes ~1-2 us, and only two waves share a SIMD, so a single step of look-ahead left the kernel
  // latency-bound (19 % MFMA utilisation in the r01 traces)
  float b2[8], b3[8];
  AStage a0 = stage_a_bf3_load<R>(apk, arr_stride, d.Mpad, 4, m0, tid);          // A(2), A(3): even / odd register set
  AStage a1 = stage_a_bf3_load<R>(apk, arr_stride, d.Mpad, 6, m0, tid);
  load_b8_bf3(d, kts, h, c, b0);
  load_b8_bf3(d, kts + 16, h, c, b1);
  load_b8_bf3(d, kts + 32, h, c, b2);
  int ks = 0;
  for (; ks + 3 < nk; ks += 4) {
    k_step_bf3<R, 0>(d, apk, arr_stride, kt4, ks, m0, c, as, kts, acc, b0, b3, a0);
    k_step_bf3<R, 1>(d, apk, arr_stride, kt4, ks + 1, m0, c, as, kts, acc, b1, b0, a1);
    k_step_bf3<R, 0>(d, apk, arr_stride, kt4, ks + 2, m0, c, as, kts, acc, b2, b1, a0);
    k_step_bf3<R, 1>(d, apk, arr_stride, kt4, ks + 3, m0, c, as, kts, acc, b3, b2, a1);
  }
  if (ks < nk) k_step_bf3<R, 0>(d, apk, arr_stride, kt4, ks, m0, c, as, kts, acc, b0, b3, a0);
  if (ks + 1 < nk) k_step_bf3<R, 1>(d, apk, arr_stride, kt4, ks + 1, m0, c, as, kts, acc, b1, b0, a1);
  if (ks + 2 < nk) k_step_bf3<R, 0>(d, apk, arr_stride, kt4, ks + 2, m0, c, as, kts, acc, b2, b1, a0);
}

template <int R, bool BF3>
__global__ __launch_bounds__(256, 2) void gemm_fwd_kernel(const FwdArgs g) {
  constexpr int BM = 32 * R;
  __shared__ __attribute__((aligned(16))) float as[BF3 ? 4 * 16 * BM : 2 * 16 * BM];   // bf16x3: 2 buffers x 2 K steps
  __shared__ __attribute__((aligned(16))) int4 kts[8 * 16];
  const rfx_gemm_desc& d = g.d;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int P = d.OA * d.OB;
  // XCD-aware tile order (block b runs on XCD b % 8, each XCD has its own L2): the channel tiles of one
  // (sample, position tile) read the same input samples, so they are made consecutive ON THE SAME XCD;
  // neighbouring position tiles are spread over the 8 XCDs.
  const int mtiles = d.Mpad / BM, ptiles = (P + 127) / 128;
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int ym = q % mtiles, pw = (q / mtiles) * 8 + xcd;      // pw: (n, position tile) work item
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

  if (BF3) run_phase_bf3<R>(d, g.apack, g.ktab, d.Kpad, m0, c, as, kts, acc);
  else run_phase<R>(d, g.apack, g.ktab, d.Kpad, m0, c, as, kts, acc);

  const rfx_epilogue& e = g.e;
  const bool two = g.apack2 != nullptr;
  // bias + activation (between the phases when there are two).  All per-row loads are issued TOGETHER, unconditionally
  // (clamped row index) and under wave-uniform tests only: with `if (e.bias) v += e.bias[m]` inside the per-element
  // loop hipcc emitted one load + s_waitcnt vmcnt(0) per element, i.e. 16*R serial L2 round trips (~40 us) per
  // workgroup -- more than the whole K loop of the short-K layers.
  {
    float bv[R][16];
    if (e.bias) {
#pragma unroll
      for (int mt = 0; mt < R; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const int mc = m < d.M ? m : d.M - 1;
          // merged phases: bias per channel; fused GLU: rows interleaved (2c, 2c+1) <-> channels (c, C+c)
          bv[mt][r] = e.bias[e.glu_out ? (mc & 1) * (d.M >> 1) + (mc >> 1) : (mc >> d.mg_log)];
        }
#pragma unroll
      for (int mt = 0; mt < R; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] += bv[mt][r];
    }
    if (e.act != RFX_ACT_NONE && !e.bwd) {
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
  if (two) {
    LaneCtx c2 = c;
    if (g.in2) {
      c2.inb = g.in2 + (c.inb - g.in);
      c2.rs = rfx_sample_rsrc(g.in2 + (int64_t)n * d.in_ns);
    }
    if (BF3) run_phase_bf3<R>(d, g.apack2, g.ktab2, g.Kpad2, m0, c2, as, kts, acc);
    else run_phase<R>(d, g.apack2, g.ktab2, g.Kpad2, m0, c2, as, kts, acc);
  }

  const int64_t opos = (int64_t)(a * d.out_sa + d.out_a0) * d.out_as + (int64_t)(b * d.out_sb + d.out_b0) * d.out_bs;
  float* outp = g.out + (int64_t)n * d.out_ns + opos;
  const float* resp = nullptr;
  if (e.res)
    resp = e.res + (int64_t)n * e.res_ns + (int64_t)(a * d.out_sa + d.out_a0) * e.res_as +
           (int64_t)(b * d.out_sb + d.out_b0) * e.res_bs;
  if (e.bwd) {
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
  if (d.mg_log) {
    // phase-merged store (see rfx_gemm_desc.mg_*): row m = channel*G + phase, position index i on the merged axis ->
    // axis index i*G + phase + mg_off.  A lane's 4 consecutive rows (r & 3) are the 4 phases of one channel when G = 4:
    // consecutive output samples, and the 32 lanes cover 32 consecutive position indices -> full lines per wave.
    const int G1 = (1 << d.mg_log) - 1;
    const int pos = d.mg_axis ? b : a;
    const int64_t other = d.mg_axis ? (int64_t)(a * d.out_sa + d.out_a0) * d.out_as : (int64_t)(b * d.out_sb + d.out_b0) * d.out_bs;
    const int64_t st = d.mg_axis ? d.out_bs : d.out_as;
    float* ob = g.out + (int64_t)n * d.out_ns + other;
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
    if (c.jvalid) {
#pragma unroll
      for (int mt = 0; mt < R; ++mt)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (m < d.M) {
            const int ch = m >> 1;
            const float a = acc[mt][r], b = acc[mt][r + 1];
            outp[(int64_t)ch * d.out_cs] = a;
            outp[(int64_t)(Ch + ch) * d.out_cs] = b;
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
  if (e.act2 != RFX_ACT_NONE) {
#pragma unroll
    for (int mt = 0; mt < R; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][r] = rfx_act_apply(acc[mt][r], e.act2, 0.f);
  }
  if (c.jvalid) {
#pragma unroll
    for (int mt = 0; mt < R; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < d.M) {
          const float v = acc[mt][r];
          outp[(int64_t)m * d.out_cs] = v;
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


// launchers of the tiled forward kernel (defined in gemm_fwd_f32.hip / gemm_fwd_bf3.hip)
int rfx_launch_gemm_fwd_f32(const FwdArgs& g, int r, dim3 grid, hipStream_t s);
int rfx_launch_gemm_fwd_bf3(const FwdArgs& g, int r, dim3 grid, hipStream_t s);
