// Shared device helpers for libremfx_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "remfx_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// bf16 STORAGE of activations (bf16 arithmetic mode): 16 stored bits per value, RNE on store, exact widening on load
struct rfx_bf16s { uint16_t v; };
// v_cvt_pk_bf16_f32 (gfx950): two round-to-nearest-even conversions per instruction.  (The software form, add 0x7fff + lsb and
// shift, cost 4 VALU instructions per VALUE; the 16-bit-storage kernels were VALU-bound on it -- r03 SQ counters)
typedef __bf16 rfx_bf16x2 __attribute__((ext_vector_type(2)));
typedef float rfx_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t rfx_cvt_pk_bf16(float lo, float hi) {
  const rfx_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, rfx_bf16x2));
}
__device__ __forceinline__ uint32_t rfx_bf16_bits(float f) { return rfx_cvt_pk_bf16(f, 0.f); }
__device__ __forceinline__ float rfx_ld1(const float* p) { return *p; }
__device__ __forceinline__ float rfx_ld1(const rfx_bf16s* p) { return __uint_as_float((uint32_t)p->v << 16); }
__device__ __forceinline__ void rfx_st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void rfx_st1(rfx_bf16s* p, float v) { p->v = (uint16_t)rfx_bf16_bits(v); }
__device__ __forceinline__ f32x4 rfx_ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 rfx_ld4(const rfx_bf16s* p) {          // 4 values = 8 bytes (8-byte aligned)
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
               __uint_as_float(u.y & 0xffff0000u)};
}
__device__ __forceinline__ void rfx_st4(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void rfx_st4(rfx_bf16s* p, const f32x4& v) {
  *reinterpret_cast<uint2*>(p) = make_uint2(rfx_cvt_pk_bf16(v[0], v[1]), rfx_cvt_pk_bf16(v[2], v[3]));
}

#define RFX_CHECK_LAUNCH()                                  \
  do {                                                      \
    hipError_t e_ = hipGetLastError();                      \
    if (e_ != hipSuccess) return -2 - (int)e_;              \
  } while (0)

// GELU by Abramowitz-Stegun 7.1.26 (|erf error| <= 1.5e-7 -- one fp32 ulp of the cdf; branch-free, one v_exp + one v_rcp); GELU and GELU'
// share the exponential, exp(-(x / sqrt2)^2) = exp(-x^2 / 2).  The library erff is ~40 instructions with branches and the
// normalise-and-activate passes were VALU-bound on it (r03 SQ counters).
__device__ __forceinline__ void rfx_gelu_parts(float x, float& cdf, float& ex) {
  const float ax = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  ex = __expf(-0.5f * x * x);
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = 1.0f - p * t * ex;                     // erf(|x| / sqrt2)
  cdf = 0.5f * (1.0f + copysignf(e, x));
}
__device__ __forceinline__ float rfx_gelu(float v) { float c, e; rfx_gelu_parts(v, c, e); return v * c; }
__device__ __forceinline__ float rfx_gelu_grad(float v) { float c, e; rfx_gelu_parts(v, c, e); return fmaf(v * 0.39894228040143267794f, e, c); }
// v_exp_f32 + v_rcp_f32 (1 ulp each): the correctly rounded fp32 division is a ~10-instruction sequence and this runs per element of
// every GLU / GLU-backward pass
__device__ __forceinline__ float rfx_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

__device__ __forceinline__ float rfx_act_apply(float v, int act, float slope) {
  switch (act) {
    case RFX_ACT_RELU: return v > 0.f ? v : 0.f;
    case RFX_ACT_GELU: return rfx_gelu(v);
    case RFX_ACT_TANH: return tanhf(v);
    case RFX_ACT_PRELU: return v >= 0.f ? v : slope * v;
    case RFX_ACT_LEAKY: return v >= 0.f ? v : 0.01f * v;
    case RFX_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    default: return v;
  }
}
// derivative wrt the pre-activation x
__device__ __forceinline__ float rfx_act_grad(float x, int act, float slope) {
  switch (act) {
    case RFX_ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case RFX_ACT_GELU: return rfx_gelu_grad(x);
    case RFX_ACT_TANH: { float t = tanhf(x); return 1.f - t * t; }
    case RFX_ACT_PRELU: return x >= 0.f ? 1.f : slope;
    case RFX_ACT_LEAKY: return x >= 0.f ? 1.f : 0.01f;
    case RFX_ACT_SIGMOID: { float s = 1.0f / (1.0f + expf(-x)); return s * (1.f - s); }
    default: return 1.f;
  }
}

// |re + i im|^2 with ONE fixed rounding sequence: the loss forward (csrc/fft.hip) and backward (csrc/losses.hip) must agree bit for bit
// on the power of a stored spectrum value (loss(x, x) has an exactly zero gradient only if |X| == |Y| there too)
__device__ __forceinline__ float rfx_pow2(float re, float im) { return __builtin_fmaf(im, im, re * re); }

__device__ __forceinline__ float rfx_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double rfx_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- deterministic two-stage sums (round 6) -----------------------------------------------------------------------------------
// A reduction over many workgroups used to end in `atomicAdd` into a zero-filled word: the result depended on the order the
// workgroups finished in (run-to-run differences in the last bits of every loss / norm / statistic), and the fill + atomics pair lost
// contributions when a second stream kept the machine busy.  Now every workgroup STORES its partial into its own
// slot of a caller-owned workspace -- slots[(row * nslots + slot) * K + k], fp64, no initialisation needed -- and
// rfx_slot_sum_kernel adds a row's slots in slot order: bit-reproducible, no fill, no atomics.
// One WAVE per output value: lane l adds slots l, l + 64, ... in order, then the fixed butterfly -- a single thread walking thousands of
// dependent loads took 60 - 120 us per launch (r06 8-clip profile).  blockDim = 256 = four outputs per workgroup.
template <typename OutT>
__global__ __launch_bounds__(256) void rfx_slot_sum_kernel(const double* __restrict__ slots, int rows, int nslots, int K, OutT* __restrict__ out) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= rows * K) return;
  const int r = i / K, k = i - r * K;
  const double* p = slots + (int64_t)r * nslots * K + k;
  double acc = 0.0;
  for (int s = lane; s < nslots; s += 64) acc += p[(int64_t)s * K];
  acc = rfx_wave_sum_d(acc);
  if (lane == 0) out[i] = (OutT)acc;
}
#define RFX_SLOT_SUM_GRID(n_outputs) dim3(((n_outputs) + 3) / 4), dim3(256)
// block-level sum of K per-thread doubles of a 256-thread workgroup into slot `slot` of row `row` (call from all threads)
template <int K>
__device__ __forceinline__ void rfx_block_store_slot(const double (&v)[K], double* __restrict__ slots, int row, int nslots, int slot) {
  __shared__ double rfx_part_[K][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const double w = rfx_wave_sum_d(v[k]);
    if (lane == 0) rfx_part_[k][wave] = w;
  }
  __syncthreads();
  if (threadIdx.x < K) {
    const int k = threadIdx.x;
    slots[((int64_t)row * nslots + slot) * K + k] = (rfx_part_[k][0] + rfx_part_[k][1]) + (rfx_part_[k][2] + rfx_part_[k][3]);
  }
}
