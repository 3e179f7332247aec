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

// ---------------------------------------------------------------------------------
// pack / unpack
// ---------------------------------------------------------------------------------
__global__ void pack_a_kernel(const float* __restrict__ w, const int32_t* __restrict__ woff,
                              int64_t w_ms, int M, int K, int Mpad, int Kpad,
                              float* __restrict__ apack) {
  const int64_t total = (int64_t)(Kpad + 32) * Mpad;   // two extra all-zero K steps (branch-free prefetch)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i / Mpad), m = (int)(i % Mpad);
    float v = 0.f;
    if (k < K && m < M) v = w[(int64_t)m * w_ms + woff[k]];
    apack[i] = v;
  }
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t bf16_rne(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// bf16x3 operand layout: two arrays (hi, lo) of [Kpad/8 + 4][Mpad] 16-byte cells, a cell = the 8
// consecutive-k bf16 values of one output row = exactly one lane's MFMA A fragment.
__global__ void pack_a_bf3_kernel(const float* __restrict__ w, const int32_t* __restrict__ woff,
                                  int64_t w_ms, int M, int K, int Mpad, int Kpad, uint4* __restrict__ apack) {
  const int64_t cells = (int64_t)(Kpad / 8 + 4) * Mpad;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int k8 = (int)(i / Mpad), m = (int)(i % Mpad);
    uint32_t hi[4] = {0, 0, 0, 0}, lo[4] = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int k = k8 * 8 + q;
      float v = 0.f;
      if (k < K && m < M) v = w[(int64_t)m * w_ms + woff[k]];
      const uint32_t h = bf16_rne(v);
      const uint32_t l = bf16_rne(v - __uint_as_float(h << 16));
      hi[q >> 1] |= h << (16 * (q & 1));
      lo[q >> 1] |= l << (16 * (q & 1));
    }
    apack[i] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    apack[cells + i] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
  }
}

__global__ void unpack_add_kernel(const float* __restrict__ dapack, const int32_t* __restrict__ woff,
                                  int64_t w_ms, int M, int K, int Kpad, float* __restrict__ dw) {
  const int64_t total = (int64_t)K * M;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / K), k = (int)(i % K);
    // distinct (k, m) map to distinct weight elements within one descriptor, but
    // several descriptors (stride phases) may run back to back on the stream.
    dw[(int64_t)m * w_ms + woff[k]] += dapack[(int64_t)m * Kpad + k];
  }
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
__device__ float rfx_zero_f32[4] = {0.f, 0.f, 0.f, 0.f};
__device__ float rfx_one_f32[4] = {1.f, 1.f, 1.f, 1.f};

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

template <int R>
__device__ __forceinline__ void k_step_bf3(const rfx_gemm_desc& d, const uint4* __restrict__ apk,
                                           int64_t arr_stride, const int4* __restrict__ kt4, int ks, int m0,
                                           const LaneCtx& c, uint4* as, int4* kts, f32x16 (&acc)[R],
                                           const float (&bc)[8], float (&bn)[8], const AStage& a_now,
                                           AStage& a_next) {
  constexpr int BM = 32 * R;
  const int tid = threadIdx.x;
  const int lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int cur = ks & 1;
  const uint4* a_lds = as + cur * 4 * BM;
  uint4 ah[R], al[R];
#pragma unroll
  for (int mt = 0; mt < R; ++mt) {
    ah[mt] = a_lds[h * BM + mt * 32 + l31];
    al[mt] = a_lds[2 * BM + h * BM + mt * 32 + l31];
  }
  // issue order matters: vmcnt retires in order and the table row + A stage are consumed (written to LDS) at
  // the end of THIS step, so they go first; the gathers, consumed RFX_BDIST steps later, go last and stay in flight
  const int4 ktreg = kt4[(ks + RFX_BDIST + 1) * 16 + (tid & 15)];
  // A tile of step ks+2 -> registers now, into LDS at the end of step ks+1 (a_now was fetched one step ago): an L2
  // round trip is longer than one K step of MFMAs
  a_next = stage_a_bf3_load<R>(apk, arr_stride, d.Mpad, 2 * (ks + 2), m0, tid);
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
  stage_a_bf3_store<R>(as + (cur ^ 1) * 4 * BM, tid, a_now);
  kts[((ks + RFX_BDIST + 1) & 7) * 16 + (tid & 15)] = ktreg;   // every thread (same value per tid & 15)
  __syncthreads();
}

template <int R>
__device__ __forceinline__ void run_phase_bf3(const rfx_gemm_desc& d, const float* __restrict__ apack,
                                              const rfx_ktab_entry* __restrict__ ktab, int Kpad, int m0,
                                              const LaneCtx& c, float* as_f, int4* kts, f32x16 (&acc)[R]) {
  const int tid = threadIdx.x;
  const int h = (tid & 63) >> 5;
  const int nk = Kpad / 16;
  if (nk == 0) return;
  const uint4* apk = reinterpret_cast<const uint4*>(apack);
  const int64_t arr_stride = (int64_t)(Kpad / 8 + 4) * d.Mpad;
  uint4* as = reinterpret_cast<uint4*>(as_f);
  const int4* kt4 = reinterpret_cast<const int4*>(ktab);
  float b0[8], b1[8];
  const AStage areg = stage_a_bf3_load<R>(apk, arr_stride, d.Mpad, 0, m0, tid);
  if (tid < 16 * (RFX_BDIST + 1)) kts[tid] = kt4[tid];   // table rows of the first K steps (table is padded)
  stage_a_bf3_store<R>(as, tid, areg);
  __syncthreads();
  // the gathers run RFX_BDIST K steps ahead of the MFMAs: one K step is ~0.2 us of matrix work, a gather that
  // misses L2 takes ~1-2 us, and only two waves share a SIMD, so a single step of look-ahead left the kernel
  // latency-bound (19 % MFMA utilisation in the r01 traces)
  float b2[8], b3[8];
  AStage a1;
  AStage a0 = stage_a_bf3_load<R>(apk, arr_stride, d.Mpad, 2, m0, tid);          // A tile of step 1
  load_b8_bf3(d, kts, h, c, b0);
  load_b8_bf3(d, kts + 16, h, c, b1);
  load_b8_bf3(d, kts + 32, h, c, b2);
  int ks = 0;
  for (; ks + 3 < nk; ks += 4) {
    k_step_bf3<R>(d, apk, arr_stride, kt4, ks, m0, c, as, kts, acc, b0, b3, a0, a1);
    k_step_bf3<R>(d, apk, arr_stride, kt4, ks + 1, m0, c, as, kts, acc, b1, b0, a1, a0);
    k_step_bf3<R>(d, apk, arr_stride, kt4, ks + 2, m0, c, as, kts, acc, b2, b1, a0, a1);
    k_step_bf3<R>(d, apk, arr_stride, kt4, ks + 3, m0, c, as, kts, acc, b3, b2, a1, a0);
  }
  if (ks < nk) k_step_bf3<R>(d, apk, arr_stride, kt4, ks, m0, c, as, kts, acc, b0, b3, a0, a1);
  if (ks + 1 < nk) k_step_bf3<R>(d, apk, arr_stride, kt4, ks + 1, m0, c, as, kts, acc, b1, b0, a1, a0);
  if (ks + 2 < nk) k_step_bf3<R>(d, apk, arr_stride, kt4, ks + 2, m0, c, as, kts, acc, b2, b1, a0, a1);
}

template <int R, bool BF3>
__global__ __launch_bounds__(256, 2) void gemm_fwd_kernel(const FwdArgs g) {
  constexpr int BM = 32 * R;
  __shared__ __attribute__((aligned(16))) float as[2 * 16 * BM];
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
  // bias + activation (between the phases when there are two)
#pragma unroll
  for (int mt = 0; mt < R; ++mt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (m < d.M) {
        float v = acc[mt][r];
        if (e.bias) v += e.bias[m];
        if (e.act != RFX_ACT_NONE && !e.bwd) {
          const float s = (e.act == RFX_ACT_PRELU) ? e.act_param[m] : 0.f;
          v = rfx_act_apply(v, e.act, s);
        }
        acc[mt][r] = v;
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
    // out = G * act'(pre);  gparam[m] += sum_j G * min(pre, 0)   (PReLU slope gradient)
#pragma unroll
    for (int mt = 0; mt < R; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        float gs = 0.f;
        if (m < d.M && c.jvalid) {
          const float pre = acc[mt][r];
          const float gin = resp[(int64_t)m * e.res_cs];
          const float s = (e.act == RFX_ACT_PRELU) ? e.act_param[m] : 0.f;
          outp[(int64_t)m * d.out_cs] = gin * rfx_act_grad(pre, e.act, s);
          gs = pre < 0.f ? gin * pre : 0.f;
        }
        if (e.gparam) {   // reduce over the 32 position lanes of this half-wave
#pragma unroll
          for (int o = 16; o > 0; o >>= 1) gs += __shfl_xor(gs, o, 64);
          if (l31 == 0 && m < d.M) atomicAdd(e.gparam + m, gs);
        }
      }
    }
    return;
  }
  float s1 = 0.f, s2 = 0.f;      // optional per-sample moments of the stored values (GroupNorm(1, C) statistics)
  if (c.jvalid) {
#pragma unroll
    for (int mt = 0; mt < R; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < d.M) {
          float v = acc[mt][r];
          if (resp) v += resp[(int64_t)m * e.res_cs];
          if (e.act2 != RFX_ACT_NONE) v = rfx_act_apply(v, e.act2, 0.f);
          outp[(int64_t)m * d.out_cs] = v;
          s1 += v; s2 += v * v;
        }
      }
    }
  }
  if (e.stat_sums) {               // wave-uniform branch: one fp64 atomic pair per wave
    const double d1 = rfx_wave_sum_d((double)s1), d2 = rfx_wave_sum_d((double)s2);
    if (lane == 0) { atomicAdd(e.stat_sums + 2 * n, d1); atomicAdd(e.stat_sums + 2 * n + 1, d2); }
  }
}

// Thin forward kernel: M <= 8 output rows (TCN output conv 256->1, tcn.py:119,129;
// last HDemucs decoders).  HBM-bound: one thread per position, K loop with
// wave-uniform weights, coalesced gathers.
template <int MM>
__device__ __forceinline__ void thin_phase(const rfx_gemm_desc& d, const rfx_ktab_entry* __restrict__ kt,
                                           const float* __restrict__ ap, int K, const float* inb,
                                           const float* safe, bool jvalid, int ia0, int ib0,
                                           float (&acc)[MM]) {
  for (int k0 = 0; k0 < K; k0 += 4) {
    float bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const rfx_ktab_entry e = kt[k0 + u];  // Kpad is a multiple of 16: always readable
      const bool ok = jvalid && (unsigned)(ia0 + e.da) < (unsigned)d.IA &&
                      (unsigned)(ib0 + e.db) < (unsigned)d.IB;
      const float* p = ok ? inb + e.off : safe;
      const float v = *p;
      bv[u] = ok ? v : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int m = 0; m < MM; ++m) acc[m] = fmaf(ap[(int64_t)(k0 + u) * d.Mpad + m], bv[u], acc[m]);
  }
}

template <int MM>
__global__ __launch_bounds__(256) void gemm_thin_fwd_kernel(const FwdArgs g) {
  const rfx_gemm_desc& d = g.d;
  const int P = d.OA * d.OB;
  const int n = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const bool jvalid = j < P;
  const int jj = jvalid ? j : 0;
  const int a = jj / d.OB, b = jj - a * d.OB;
  const int ia0 = a * d.SA, ib0 = b * d.SB;
  const int64_t ioff = (int64_t)n * d.in_ns + (int64_t)ia0 * d.in_as + (int64_t)ib0 * d.in_bs;
  float acc[MM];
#pragma unroll
  for (int m = 0; m < MM; ++m) acc[m] = 0.f;
  thin_phase<MM>(d, g.ktab, g.apack, d.K, g.in + ioff, g.in, jvalid, ia0, ib0, acc);
  const rfx_epilogue& e = g.e;
#pragma unroll
  for (int m = 0; m < MM; ++m) {
    if (m < d.M) {
      float v = acc[m];
      if (e.bias) v += e.bias[m];
      if (e.act != RFX_ACT_NONE && !e.bwd)
        v = rfx_act_apply(v, e.act, e.act == RFX_ACT_PRELU ? e.act_param[m] : 0.f);
      acc[m] = v;
    }
  }
  if (g.apack2) {
    const float* in2 = g.in2 ? g.in2 : g.in;
    thin_phase<MM>(d, g.ktab2, g.apack2, g.Kpad2, in2 + ioff, in2, jvalid, ia0, ib0, acc);
  }
  if (!jvalid) return;
  float st1 = 0.f, st2 = 0.f;
  const int64_t opos = (int64_t)(a * d.out_sa + d.out_a0) * d.out_as + (int64_t)(b * d.out_sb + d.out_b0) * d.out_bs;
  float* outp = g.out + (int64_t)n * d.out_ns + opos;
#pragma unroll
  for (int m = 0; m < MM; ++m) {
    if (m < d.M) {
      float v = acc[m];
      float r = 0.f;
      if (e.res)
        r = e.res[(int64_t)n * e.res_ns + (int64_t)m * e.res_cs +
                  (int64_t)(a * d.out_sa + d.out_a0) * e.res_as + (int64_t)(b * d.out_sb + d.out_b0) * e.res_bs];
      if (e.bwd) {
        const float s = (e.act == RFX_ACT_PRELU) ? e.act_param[m] : 0.f;
        outp[(int64_t)m * d.out_cs] = r * rfx_act_grad(v, e.act, s);
        if (e.gparam && v < 0.f) atomicAdd(e.gparam + m, r * v);
      } else {
        v += r;
        if (e.act2 != RFX_ACT_NONE) v = rfx_act_apply(v, e.act2, 0.f);
        outp[(int64_t)m * d.out_cs] = v;
        st1 += v; st2 += v * v;
      }
    }
  }
  if (e.stat_sums) {
    atomicAdd(e.stat_sums + 2 * n, (double)st1);       // thin path: tiny tensors, per-thread atomics are fine
    atomicAdd(e.stat_sums + 2 * n + 1, (double)st2);
  }
}

// ---------------------------------------------------------------------------------
// weight-gradient MFMA kernel:  dapack[k][m] += sum_p g[m][p] * In(k, p)
// Both operands are contiguous along the reduction axis p in memory, so both are
// staged through LDS ([row][32 positions], stride 33 -> conflict-free operand reads).
// ---------------------------------------------------------------------------------
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
  int xcd_grouped;       // 1: 1-D grid, all (k, m) tiles of one position split share an XCD (ids congruent mod 8)
};

template <int TM, int TK>
__global__ __launch_bounds__(256) void gemm_wgrad_kernel(const WgradArgs w) {
  constexpr int RM = 64 * TM, RK = 64 * TK, LD = 33;
  __shared__ float gs[RM * LD];
  __shared__ float xs[RK * LD];
  __shared__ rfx_ktab_entry kts[RK];
  const rfx_gemm_desc& d = w.d;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wk = wave & 1;
  // plain order (k tile fastest).  An XCD-grouped order (all tiles of one position split on one XCD) was
  // measured SLOWER (132 -> 116 TF/s-eq at the TCN shape): the splits are too few / too coarse to balance.
  const int zsplit = blockIdx.z;
  const int m0 = blockIdx.y * RM;
  const int k0 = blockIdx.x * RK;
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
  const int prow = tid >> 5;  // 0..7: row group for loads; lane position = tid & 31
  const int pl = tid & 31;
  float gv[RM / 8], xv[RK / 8];
  // Gathers of one 32-position tile into registers.  Invalid lanes read a device 0 (or 1 for the
  // bias column) instead of masking the loaded value, so nothing consumes the result until the
  // LDS store of the NEXT iteration: the loads stay in flight under this tile's MFMAs.
  auto load_tile = [&](int t) {
    const int n = t / w.tiles_per_sample;
    const int j = (t - n * w.tiles_per_sample) * 32 + pl;
    const bool jvalid = j < P;
    const int jj = jvalid ? j : 0;
    const int a = jj / d.OB, b = jj - a * d.OB;
    const int ia0 = a * d.SA, ib0 = b * d.SB;
    const float* inb = w.in + (int64_t)n * d.in_ns + (int64_t)ia0 * d.in_as + (int64_t)ib0 * d.in_bs;
    const float* gb = w.g + (int64_t)n * d.out_ns + (int64_t)(a * d.out_sa + d.out_a0) * d.out_as +
                      (int64_t)(b * d.out_sb + d.out_b0) * d.out_bs;
#pragma unroll
    for (int i = 0; i < RM / 8; ++i) {
      const int m = m0 + prow + 8 * i;
      const bool ok = jvalid & (m < d.M);
      const float* p = ok ? gb + (int64_t)m * d.out_cs : rfx_zero_f32;
      gv[i] = *p;
    }
#pragma unroll
    for (int i = 0; i < RK / 8; ++i) {
      const rfx_ktab_entry e = kts[prow + 8 * i];
      const bool ones = e.flags & 1;
      const bool ok = jvalid & !ones & ((unsigned)(ia0 + e.da) < (unsigned)d.IA) &
                      ((unsigned)(ib0 + e.db) < (unsigned)d.IB);
      const float* p = ok ? inb + e.off : ((ones & jvalid) ? rfx_one_f32 : rfx_zero_f32);
      xv[i] = *p;
    }
  };
  if (t_begin < t_end) load_tile(t_begin);
  for (int t = t_begin; t < t_end; ++t) {
    __syncthreads();  // previous tile's operand reads are done
#pragma unroll
    for (int i = 0; i < RM / 8; ++i) gs[(prow + 8 * i) * LD + pl] = gv[i];
#pragma unroll
    for (int i = 0; i < RK / 8; ++i) xs[(prow + 8 * i) * LD + pl] = xv[i];
    __syncthreads();
    load_tile(t + 1 < t_end ? t + 1 : t);   // unconditional (branch-free): the last tile is re-read
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float av[TM], bv[TK];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) av[tm] = gs[(wm * 32 * TM + tm * 32 + l31) * LD + 2 * kk + h];
#pragma unroll
      for (int tk = 0; tk < TK; ++tk) bv[tk] = xs[(wk * 32 * TK + tk * 32 + l31) * LD + 2 * kk + h];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tk = 0; tk < TK; ++tk)
          acc[tm][tk] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm], bv[tk], acc[tm][tk], 0, 0, 0);
    }
  }
  // D[i = m][j = k] -> dapack[m][k]  (k = lane axis -> 128-byte coalesced atomics)
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tk = 0; tk < TK; ++tk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 * TM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int k = k0 + wk * 32 * TK + tk * 32 + l31;
        if (m < d.M && k < d.K) atomicAdd(w.dapack + (int64_t)m * d.Kpad + k, acc[tm][tk][r]);
      }
}

// bf16x3 weight gradient: same tiling and gathers as gemm_wgrad_kernel, but the two LDS tiles hold
// the operands pre-split into bf16 hi / lo halves ([row][32 positions], 80-byte rows: 16-byte aligned
// MFMA fragments, conflict-free ds_read_b128) and the product runs on v_mfma_f32_32x32x16_bf16.
template <int TM, int TK>
__global__ __launch_bounds__(256) void gemm_wgrad_bf3_kernel(const WgradArgs w) {
  constexpr int RM = 64 * TM, RK = 64 * TK, LDW = 40;   // bf16 elements per LDS row
  __shared__ __attribute__((aligned(16))) unsigned short gs_hi[RM * LDW], gs_lo[RM * LDW];
  __shared__ __attribute__((aligned(16))) unsigned short xs_hi[RK * LDW], xs_lo[RK * LDW];
  __shared__ rfx_ktab_entry kts[RK];
  const rfx_gemm_desc& d = w.d;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wk = wave & 1;
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
  auto load_tile = [&](int t, Stage& st) {
    const int n = t / w.tiles_per_sample;                       // wave-uniform
    const int j = (t - n * w.tiles_per_sample) * 32 + pl;
    const bool jvalid = j < P;
    const int jj = jvalid ? j : 0;
    const int a = jj / d.OB, b = jj - a * d.OB;
    const int ia0 = a * d.SA, ib0 = b * d.SB;
    const __amdgpu_buffer_rsrc_t irs = rfx_sample_rsrc(w.in + (int64_t)n * d.in_ns);
    const __amdgpu_buffer_rsrc_t grs = rfx_sample_rsrc(w.g + (int64_t)n * d.out_ns);
    const uint32_t voff = (uint32_t)(((int64_t)ia0 * d.in_as + (int64_t)ib0 * d.in_bs) * 4);
    const uint32_t goff = (uint32_t)(((int64_t)(a * d.out_sa + d.out_a0) * d.out_as +
                                      (int64_t)(b * d.out_sb + d.out_b0) * d.out_bs + (int64_t)(m0 + prow) * d.out_cs) * 4);
    const uint32_t gstep = (uint32_t)(8 * d.out_cs * 4);
    st.jv = jvalid ? 1.f : 0.f;
#pragma unroll
    for (int i = 0; i < RM / 8; ++i) {
      const bool ok = jvalid & (m0 + prow + 8 * i < d.M);
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
    const __bf16 l = (__bf16)(v - __uint_as_float((uint32_t)hb << 16));
    hi[row * LDW + pl] = hb;
    lo[row * LDW + pl] = __builtin_bit_cast(unsigned short, l);
  };
  auto stage = [&](const Stage& st) {
#pragma unroll
    for (int i = 0; i < RM / 8; ++i) put(gs_hi, gs_lo, prow + 8 * i, st.gv[i]);
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
        al[tm] = *reinterpret_cast<const bf16x8*>(gs_lo + off);
      }
#pragma unroll
      for (int tk = 0; tk < TK; ++tk) {
        const int off = (wk * 32 * TK + tk * 32 + l31) * LDW + 16 * ks2 + 8 * h;
        bh[tk] = *reinterpret_cast<const bf16x8*>(xs_hi + off);
        bl[tk] = *reinterpret_cast<const bf16x8*>(xs_lo + off);
      }
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tk = 0; tk < TK; ++tk) {
          acc[tm][tk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bh[tk], acc[tm][tk], 0, 0, 0);
          acc[tm][tk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bl[tk], acc[tm][tk], 0, 0, 0);
          acc[tm][tk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[tm], bh[tk], acc[tm][tk], 0, 0, 0);
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
        if (m < d.M && k < d.K) atomicAdd(w.dapack + (int64_t)m * d.Kpad + k, acc[tm][tk][r]);
      }
}

// Thin weight gradient (M <= 8): one wave per k row, lanes along positions.
template <int MM>
__global__ __launch_bounds__(256) void gemm_thin_wgrad_kernel(const WgradArgs w) {
  const rfx_gemm_desc& d = w.d;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int k = blockIdx.x * 4 + wave;
  if (k >= d.K) return;
  const rfx_ktab_entry e = w.ktab[k];
  const bool ones = e.flags & 1;
  const int P = d.OA * d.OB;
  float acc[MM];
#pragma unroll
  for (int m = 0; m < MM; ++m) acc[m] = 0.f;
  const int64_t total = (int64_t)d.N * P;
  const int64_t chunk = (total + gridDim.y - 1) / gridDim.y;
  const int64_t q0 = (int64_t)blockIdx.y * chunk;
  const int64_t q1 = min(q0 + chunk, total);
  for (int64_t q = q0 + lane; q < q1; q += 64) {
    const int n = (int)(q / P);
    const int j = (int)(q - (int64_t)n * P);
    const int a = j / d.OB, b = j - a * d.OB;
    const int ia0 = a * d.SA, ib0 = b * d.SB;
    float xv;
    if (ones) xv = 1.f;
    else {
      const bool ok = (unsigned)(ia0 + e.da) < (unsigned)d.IA && (unsigned)(ib0 + e.db) < (unsigned)d.IB;
      xv = ok ? w.in[(int64_t)n * d.in_ns + (int64_t)ia0 * d.in_as + (int64_t)ib0 * d.in_bs + e.off] : 0.f;
    }
    const float* gb = w.g + (int64_t)n * d.out_ns + (int64_t)(a * d.out_sa + d.out_a0) * d.out_as +
                      (int64_t)(b * d.out_sb + d.out_b0) * d.out_bs;
#pragma unroll
    for (int m = 0; m < MM; ++m)
      if (m < d.M) acc[m] = fmaf(gb[(int64_t)m * d.out_cs], xv, acc[m]);
  }
#pragma unroll
  for (int m = 0; m < MM; ++m) {
    const float s = rfx_wave_sum(acc[m]);
    if (lane == 0 && m < d.M) atomicAdd(w.dapack + (int64_t)m * d.Kpad + k, s);
  }
}

// ---------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------
static bool desc_ok(const rfx_gemm_desc* d) {
  return d && d->N > 0 && d->M > 0 && d->K >= 0 && d->OA > 0 && d->OB > 0 && d->Mpad % 4 == 0 &&
         d->Kpad % 16 == 0 && d->Kpad >= d->K && d->Mpad >= d->M;
}

extern "C" int rfx_abi_version(void) { return RFX_ABI_VERSION; }

extern "C" int rfx_pack_a(const float* w, const int32_t* woff, int64_t w_ms, int32_t M, int32_t K,
                          int32_t Mpad, int32_t Kpad, int32_t prec, float* apack, void* stream) {
  if (!w || !woff || !apack || M <= 0 || K < 0 || Mpad < M || Kpad < K) return -1;
  if (prec == 1) {
    const int64_t cells = (int64_t)(Kpad / 8 + 4) * Mpad;
    const int grid = (int)((cells + 255) / 256 < 4096 ? (cells + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_a_bf3_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, woff, w_ms, M, K, Mpad,
                       Kpad, reinterpret_cast<uint4*>(apack));
    RFX_CHECK_LAUNCH();
    return 0;
  }
  const int64_t total = (int64_t)(Kpad + 32) * Mpad;
  if (total == 0) return 0;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_a_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, woff, w_ms, M, K,
                     Mpad, Kpad, apack);
  RFX_CHECK_LAUNCH();
  return 0;
}

extern "C" int rfx_unpack_add(const float* dapack, const int32_t* woff, int64_t w_ms, int32_t M,
                              int32_t K, int32_t Kpad, float* dw, void* stream) {
  if (!dapack || !woff || !dw || M <= 0 || K < 0 || Kpad < K) return -1;
  const int64_t total = (int64_t)K * M;
  if (total == 0) return 0;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(unpack_add_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dapack, woff,
                     w_ms, M, K, Kpad, dw);
  RFX_CHECK_LAUNCH();
  return 0;
}

// R (channel tiles per wave) is a pure function of M so that host-side packing
// and the kernel agree on Mpad = ceil(M / 32R) * 32R.
static int pick_r(int M, int K) {
  if (M <= 8) return 0;  // thin path
  if (M <= 32) return 1;
  // short reductions are output-write bound: one 32-row tile per wave keeps the kernel at ~100 VGPRs
  // (4 waves per SIMD instead of 2) so more stores / gathers are in flight per CU
  if (K <= 64) return 1;
  int best = 4, best_pad = ((M + 127) / 128) * 128;
  for (int r = 3; r >= 2; --r) {
    const int bm = 32 * r, pad = ((M + bm - 1) / bm) * bm;
    if (pad < best_pad) { best = r; best_pad = pad; }
  }
  return best;
}

extern "C" int rfx_gemm_pick_r(int32_t M, int32_t K) { return pick_r(M, K); }

extern "C" int rfx_gemm_fwd(const rfx_gemm_desc* d, const float* apack, const rfx_ktab_entry* ktab,
                            const float* in, float* out, const rfx_epilogue* epi, const float* apack2,
                            const rfx_ktab_entry* ktab2, int32_t K2, int32_t Kpad2, const float* in2,
                            int32_t prec, void* stream) {
  if (!desc_ok(d) || !apack || !ktab || !in || !out) return -1;
  if ((apack2 != nullptr) != (ktab2 != nullptr)) return -1;
  if (apack2 && (Kpad2 % 16 != 0 || Kpad2 < K2)) return -1;
  FwdArgs g;
  g.d = *d;
  g.apack = apack; g.ktab = ktab; g.in = in; g.out = out;
  if (epi) g.e = *epi;
  else { g.e = rfx_epilogue{}; }
  if (g.e.bwd && !g.e.res) return -1;
  g.apack2 = apack2; g.ktab2 = ktab2; g.Kpad2 = apack2 ? Kpad2 : 0; g.in2 = in2;
  const int P = d->OA * d->OB;
  const int r = d->R;
  if (r < 0 || r > 4 || (r == 0) != (d->M <= 8)) return -1;
  hipStream_t s = (hipStream_t)stream;
  if (r == 0) {
    dim3 grid((P + 255) / 256, d->N);
    if (d->M <= 1) hipLaunchKernelGGL(gemm_thin_fwd_kernel<1>, grid, dim3(256), 0, s, g);
    else if (d->M <= 2) hipLaunchKernelGGL(gemm_thin_fwd_kernel<2>, grid, dim3(256), 0, s, g);
    else if (d->M <= 4) hipLaunchKernelGGL(gemm_thin_fwd_kernel<4>, grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL(gemm_thin_fwd_kernel<8>, grid, dim3(256), 0, s, g);
    RFX_CHECK_LAUNCH();
    return 0;
  }
  const int bm = 32 * r;
  if (d->Mpad % bm != 0) return -1;
  const int64_t work = (int64_t)((P + 127) / 128) * d->N;          // (sample, position tile) items
  const int64_t nblk = ((work + 7) / 8) * 8 * (d->Mpad / bm);
  if (nblk > 0x7fffffff) return -1;
  dim3 grid((unsigned)nblk);
  if (prec == 1) {
    switch (r) {
      case 1: hipLaunchKernelGGL((gemm_fwd_kernel<1, true>), grid, dim3(256), 0, s, g); break;
      case 2: hipLaunchKernelGGL((gemm_fwd_kernel<2, true>), grid, dim3(256), 0, s, g); break;
      case 3: hipLaunchKernelGGL((gemm_fwd_kernel<3, true>), grid, dim3(256), 0, s, g); break;
      default: hipLaunchKernelGGL((gemm_fwd_kernel<4, true>), grid, dim3(256), 0, s, g); break;
    }
  } else {
    switch (r) {
      case 1: hipLaunchKernelGGL((gemm_fwd_kernel<1, false>), grid, dim3(256), 0, s, g); break;
      case 2: hipLaunchKernelGGL((gemm_fwd_kernel<2, false>), grid, dim3(256), 0, s, g); break;
      case 3: hipLaunchKernelGGL((gemm_fwd_kernel<3, false>), grid, dim3(256), 0, s, g); break;
      default: hipLaunchKernelGGL((gemm_fwd_kernel<4, false>), grid, dim3(256), 0, s, g); break;
    }
  }
  RFX_CHECK_LAUNCH();
  return 0;
}

extern "C" int rfx_gemm_wgrad(const rfx_gemm_desc* d, const rfx_ktab_entry* ktab, const float* in,
                              const float* gout, float* dapack, int32_t prec, void* stream) {
  if (!desc_ok(d) || !ktab || !in || !gout || !dapack) return -1;
  if (d->K == 0) return 0;
  WgradArgs w;
  w.d = *d; w.ktab = ktab; w.in = in; w.g = gout; w.dapack = dapack;
  const int P = d->OA * d->OB;
  w.tiles_per_sample = (P + 31) / 32;
  w.total_tiles = d->N * w.tiles_per_sample;
  hipStream_t s = (hipStream_t)stream;
  if (d->M <= 8) {
    const int64_t total = (int64_t)d->N * P;
    int splits = (int)(total / 4096 < 1 ? 1 : (total / 4096 > 64 ? 64 : total / 4096));
    dim3 grid((d->K + 3) / 4, splits);
    w.tiles_per_block = 0;
    if (d->M <= 1) hipLaunchKernelGGL(gemm_thin_wgrad_kernel<1>, grid, dim3(256), 0, s, w);
    else if (d->M <= 2) hipLaunchKernelGGL(gemm_thin_wgrad_kernel<2>, grid, dim3(256), 0, s, w);
    else if (d->M <= 4) hipLaunchKernelGGL(gemm_thin_wgrad_kernel<4>, grid, dim3(256), 0, s, w);
    else hipLaunchKernelGGL(gemm_thin_wgrad_kernel<8>, grid, dim3(256), 0, s, w);
    RFX_CHECK_LAUNCH();
    return 0;
  }
  const int tm = d->M > 64 ? 2 : 1;
  const int tk = d->K > 64 ? 2 : 1;
  const int mt = (d->M + 64 * tm - 1) / (64 * tm), kt = (d->K + 64 * tk - 1) / (64 * tk);
  // aim for ~2048 workgroups; each should still see >= 16 position tiles
  int splits = max(1, 2048 / (mt * kt));
  splits = min(splits, max(1, w.total_tiles / 64));
  w.tiles_per_block = (w.total_tiles + splits - 1) / splits;
  splits = (w.total_tiles + w.tiles_per_block - 1) / w.tiles_per_block;
  w.kt = kt; w.mt = mt; w.splits = splits;
  dim3 grid(kt, mt, splits);
  w.xcd_grouped = 0;
  if (prec == 1) {
    static const int xcd_mode = getenv("RFX_WGRAD_XCD") ? atoi(getenv("RFX_WGRAD_XCD")) : 0;   // measured on Demucs B=64: 408.0 ms off, 411.8 ms on
    if (xcd_mode > 0 && splits >= xcd_mode) {
      w.xcd_grouped = 1;
      grid = dim3(((splits + 7) / 8) * 8 * kt * mt, 1, 1);
    }
    if (tm == 2 && tk == 2) hipLaunchKernelGGL((gemm_wgrad_bf3_kernel<2, 2>), grid, dim3(256), 0, s, w);
    else if (tm == 2) hipLaunchKernelGGL((gemm_wgrad_bf3_kernel<2, 1>), grid, dim3(256), 0, s, w);
    else if (tk == 2) hipLaunchKernelGGL((gemm_wgrad_bf3_kernel<1, 2>), grid, dim3(256), 0, s, w);
    else hipLaunchKernelGGL((gemm_wgrad_bf3_kernel<1, 1>), grid, dim3(256), 0, s, w);
    RFX_CHECK_LAUNCH();
    return 0;
  }
  if (tm == 2 && tk == 2) hipLaunchKernelGGL((gemm_wgrad_kernel<2, 2>), grid, dim3(256), 0, s, w);
  else if (tm == 2) hipLaunchKernelGGL((gemm_wgrad_kernel<2, 1>), grid, dim3(256), 0, s, w);
  else if (tk == 2) hipLaunchKernelGGL((gemm_wgrad_kernel<1, 2>), grid, dim3(256), 0, s, w);
  else hipLaunchKernelGGL((gemm_wgrad_kernel<1, 1>), grid, dim3(256), 0, s, w);
  RFX_CHECK_LAUNCH();
  return 0;
}
