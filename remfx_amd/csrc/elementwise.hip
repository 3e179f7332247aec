// Elementwise activations and small reductions (HBM-bound, float4 grid-stride).
// Replaces F.gelu / F.relu / torch.tanh / nn.PReLU backward / nn.L1Loss call
// sites of the hot path (tcn.py:51,129; models.py:320; HDemucs enc/dec).
#include "common.h"
#include <type_traits>

__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, int act) {
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = rfx_act_apply(v[c], act, 0.f);
    reinterpret_cast<f32x4*>(y)[i] = v;
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    y[i] = rfx_act_apply(x[i], act, 0.f);
}

// y = act(x) + res: the decoder's GELU and the NEXT layer's skip-connection add in one pass (3 tensor passes instead of 5)
__global__ void act_add_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res, float* __restrict__ y, int64_t n,
                                   int act) {
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    const f32x4 r = reinterpret_cast<const f32x4*>(res)[i];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = rfx_act_apply(v[c], act, 0.f) + r[c];
    reinterpret_cast<f32x4*>(y)[i] = v;
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    y[i] = rfx_act_apply(x[i], act, 0.f) + res[i];
}

__global__ void act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                               float* __restrict__ gx, int64_t n, int act) {
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    f32x4 g = reinterpret_cast<const f32x4*>(gy)[i];
#pragma unroll
    for (int c = 0; c < 4; ++c) g[c] *= rfx_act_grad(v[c], act, 0.f);
    reinterpret_cast<f32x4*>(gx)[i] = g;
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    gx[i] = gy[i] * rfx_act_grad(x[i], act, 0.f);
}

// x, gy: [N][C][L] contiguous; one block per (n-chunk, c) row set.
__global__ void prelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                 const float* __restrict__ slope, float* __restrict__ gx,
                                 double* __restrict__ slots /* [C][N] */, int64_t N, int64_t C, int64_t L) {
  const int64_t row = blockIdx.x;  // n*C + c
  const int c = (int)(row % C);
  const float s = slope[c];
  const float* xr = x + row * L;
  const float* gr = gy + row * L;
  float* gxr = gx + row * L;
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < L; i += blockDim.x) {
    const float v = xr[i], g = gr[i];
    gxr[i] = v >= 0.f ? g : s * g;
    acc += v >= 0.f ? 0.f : g * v;
  }
  acc = rfx_wave_sum(acc);
  __shared__ float part[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) part[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) slots[(int64_t)c * N + row / C] = (double)((part[0] + part[1]) + (part[2] + part[3]));
}

__global__ __launch_bounds__(256) void l1_sum_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                                                     double* __restrict__ slots) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  double acc[1] = {0.0};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    acc[0] += (double)fabsf(a[i] - b[i]);
  rfx_block_store_slot<1>(acc, slots, 0, gridDim.x, blockIdx.x);
}

__global__ __launch_bounds__(64) void l1_finish_kernel(const double* __restrict__ slots, int nslots, float scale, float* __restrict__ out) {
  double acc = 0.0;                     // one wave: lane l adds slots l, l + 64, ... in order, then the fixed butterfly
  for (int k = threadIdx.x; k < nslots; k += 64) acc += slots[k];
  acc = rfx_wave_sum_d(acc);
  if (threadIdx.x == 0) out[0] = (float)acc * scale;          // torch: (fp32 sum) / numel
}

// slots[c][chunk] = partial sum_{n,a,b} x[...] of the chunk; grid = (C, chunks); added in chunk order by rfx_slot_sum_kernel
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ x, int N, int Cn, int A, int B, int64_t ns,
                                                          int64_t cs, int64_t as, int64_t bs, double* __restrict__ slots) {
  const int c = blockIdx.x;
  const int64_t per = (int64_t)A * B, total = (int64_t)N * per;
  const int64_t chunk = (total + gridDim.y - 1) / gridDim.y;
  const int64_t q0 = (int64_t)blockIdx.y * chunk, q1 = q0 + chunk < total ? q0 + chunk : total;
  double acc = 0.0;
  // (n, r) advance incrementally: the two 64-bit divisions per element of the first version made this reduction
  // VALU-bound (1.2 TB/s); the common contiguous (a, b) plane needs no (a, b) split at all
  const bool plane = (bs == 1 && as == (int64_t)B);
  int64_t q = q0 + threadIdx.x;
  int64_t n = q / per, r = q - n * per;
  const float* xc = x + (int64_t)c * cs;
  for (; q < q1; q += blockDim.x) {
    int64_t off = n * ns;
    if (plane) off += r;
    else { const int a = (int)((uint32_t)r / (uint32_t)B); off += (int64_t)a * as + (int64_t)((int)r - a * B) * bs; }
    acc += (double)xc[off];
    r += blockDim.x;
    while (r >= per) { r -= per; ++n; }
  }
  const double v[1] = {acc};
  rfx_block_store_slot<1>(v, slots, c, gridDim.y, blockIdx.y);
}

// out[n,c,a,b] = x[n,c,a,b] + alpha * y[n*yn + c*yc + a*ya + b*yb]   (x, out contiguous; y strided, stride 0 =
// broadcast): skip / inject / frequency-embedding adds of HDemucs (torchaudio HDemucs via models.py:319).
__global__ void add_bcast_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out,
                                 int64_t total, int C, int A, int B, int64_t yn, int64_t yc, int64_t ya, int64_t yb,
                                 float alpha) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i % B);
    int64_t r = i / B;
    const int a = (int)(r % A); r /= A;
    const int c = (int)(r % C);
    const int64_t n = r / C;
    out[i] = x[i] + alpha * y[n * yn + c * yc + a * ya + b * yb];
  }
}

// per-row mean and UNBIASED std of x[R][L] (fp64 accumulation): HDemucs input / spectrogram standardisation.
// Every workgroup stores its pair into its own slot (sums[r][slot][2]); the finalize kernel adds a row's slots in order.
__global__ __launch_bounds__(256) void row_moments_kernel(const float* __restrict__ x, int64_t L, double* __restrict__ sums) {
  const int r = blockIdx.y;
  const float* xr = x + (int64_t)r * L;
  double v[2] = {0.0, 0.0};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < L; i += (int64_t)gridDim.x * 256) {
    const double t = xr[i];
    v[0] += t; v[1] += t * t;
  }
  rfx_block_store_slot<2>(v, sums, r, gridDim.x, blockIdx.x);
}
// one wave per row: lane l adds the row's slots l, l + 64, ... in order, then the fixed butterfly
__global__ __launch_bounds__(256) void row_moments_finalize_kernel(const double* __restrict__ sums, int R, int nslots, double L, float* __restrict__ mean,
                                                                   float* __restrict__ stdv, float eps, float* __restrict__ coef_a,
                                                                   float* __restrict__ coef_b) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= R) return;
  double s1 = 0.0, s2 = 0.0;
  for (int k = lane; k < nslots; k += 64) { s1 += sums[2 * ((int64_t)r * nslots + k)]; s2 += sums[2 * ((int64_t)r * nslots + k) + 1]; }
  s1 = rfx_wave_sum_d(s1); s2 = rfx_wave_sum_d(s2);
  if (lane != 0) return;
  const double m = s1 / L;
  double var = (s2 - L * m * m) / (L - 1.0);
  var = var > 0.0 ? var : 0.0;
  const float mf = (float)m, sf = (float)sqrt(var);
  mean[r] = mf;
  stdv[r] = sf;
  if (coef_a) {                        // the standardisation as one affine map: y = x a + b, a = 1 / (eps + std), b = -mean a
    const float a = 1.0f / (eps + sf);
    coef_a[r] = a;
    coef_b[r] = -mf * a;
  }
}
// out[r][i] = x[r][i] * a[r] + b[r]
__global__ void row_affine_kernel(const float* __restrict__ x, const float* __restrict__ a, const float* __restrict__ b,
                                  float* __restrict__ out, int64_t L, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / L;
    out[i] = x[i] * a[r] + (b ? b[r] : 0.f);
  }
}

static int grid_for(int64_t n) {
  const int64_t b = (n + 1023) / 1024;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

extern "C" int rfx_act_fwd(const float* x, float* y, int64_t n, int32_t act, void* stream) {
  if (!x || !y || n < 0) return -1;
  if (n == 0) return 0;
  hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, act);
  RFX_CHECK_LAUNCH();
  return 0;
}
// Activation between two row layouts: the tensor is a (D0, D1, D2) grid of contiguous T-float rows whose row offsets
// differ between x, gy and out (HDemucs freq branch: GELU output written as (B, Fr, C, T) so that the DConv's
// (B*Fr, C, T) view is free, and its backward reading the gradient from that layout).  gy == nullptr: out = act(x);
// otherwise out = gy * act'(x).  One wave per (row, 256-sample chunk), wave-uniform 32-bit decode.
struct ActRowsArgs {
  const float* x; const float* gy; float* out;
  int64_t xs[3], gs[3], os[3];
  int D1, D2, T, act;
  uint32_t nitems, ipr;
};
// X16: x is STORED as bf16 (a conv output of the bf16 arithmetic mode, ops.bf16_storage) -- and so is the gradient this kernel
// returns for it (autograd hands a 16-bit tensor a 16-bit gradient; its readers are GEMM operands, which round to bf16 anyway).
// The forward result stays fp32.  Strides of x / the backward output are in ELEMENTS of their storage type.
template <bool VEC, bool X16>
__global__ __launch_bounds__(256) void act_rows_kernel(const ActRowsArgs a) {
  typedef typename std::conditional<X16, rfx_bf16s, float>::type XT;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t w = blockIdx.x * 4u + wave;
  if (w >= a.nitems) return;
  const uint32_t row = w / a.ipr, ck = w - row * a.ipr;
  const uint32_t r1 = row / (uint32_t)a.D2, i2 = row - r1 * (uint32_t)a.D2;
  const uint32_t i0 = r1 / (uint32_t)a.D1, i1 = r1 - i0 * (uint32_t)a.D1;
  const XT* xr = reinterpret_cast<const XT*>(a.x) + (int64_t)i0 * a.xs[0] + (int64_t)i1 * a.xs[1] + (int64_t)i2 * a.xs[2];
  const float* gr = a.gy ? a.gy + (int64_t)i0 * a.gs[0] + (int64_t)i1 * a.gs[1] + (int64_t)i2 * a.gs[2] : nullptr;
  const int64_t ooff = (int64_t)i0 * a.os[0] + (int64_t)i1 * a.os[1] + (int64_t)i2 * a.os[2];
  float* orow = a.out + ooff;                                        // forward (and fp32 backward) output
  XT* orow_x = reinterpret_cast<XT*>(a.out) + ooff;                  // backward output in x's storage type
  const int lane = threadIdx.x & 63;
  if (VEC) {
    const int t = (int)ck * 256 + lane * 4;
    if (t >= a.T) return;
    f32x4 v = rfx_ld4(xr + t);
    if (gr) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gr + t);
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = g[c] * rfx_act_grad(v[c], a.act, 0.f);
      rfx_st4(orow_x + t, v);
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = rfx_act_apply(v[c], a.act, 0.f);
      *reinterpret_cast<f32x4*>(orow + t) = v;
    }
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int t = (int)ck * 256 + q * 64 + lane;
      if (t < a.T) {
        if (gr) rfx_st1(orow_x + t, gr[t] * rfx_act_grad(rfx_ld1(xr + t), a.act, 0.f));
        else orow[t] = rfx_act_apply(rfx_ld1(xr + t), a.act, 0.f);
      }
    }
  }
}
static int act_rows_launch(bool x16, const void* x, int64_t xs0, int64_t xs1, int64_t xs2, const float* gy, int64_t gs0,
                           int64_t gs1, int64_t gs2, void* out, int64_t os0, int64_t os1, int64_t os2, int32_t D0,
                           int32_t D1, int32_t D2, int32_t T, int32_t act, void* stream) {
  if (!x || !out || D0 <= 0 || D1 <= 0 || D2 <= 0 || T <= 0) return -1;
  const int64_t rows = (int64_t)D0 * D1 * D2, per = ((int64_t)T + 255) / 256;
  if (rows * per > 0x7fffffffLL) return -1;
  ActRowsArgs a;
  a.x = static_cast<const float*>(x); a.gy = gy; a.out = static_cast<float*>(out);
  a.xs[0] = xs0; a.xs[1] = xs1; a.xs[2] = xs2; a.gs[0] = gs0; a.gs[1] = gs1; a.gs[2] = gs2;
  a.os[0] = os0; a.os[1] = os1; a.os[2] = os2;
  a.D1 = D1; a.D2 = D2; a.T = T; a.act = act; a.nitems = (uint32_t)(rows * per); a.ipr = (uint32_t)per;
  const int xal = x16 ? 7 : 15, oal = (x16 && gy) ? 7 : 15;        // 4 elements of the storage type
  const bool vec = !(T & 3) && !((xs0 | xs1 | xs2 | os0 | os1 | os2 | (gy ? (gs0 | gs1 | gs2) : 0)) & 3) &&
                   !((uintptr_t)x & xal) && !((uintptr_t)out & oal) && !((uintptr_t)gy & 15);
  const dim3 grid((a.nitems + 3) / 4);
  if (x16) {
    if (vec) hipLaunchKernelGGL((act_rows_kernel<true, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((act_rows_kernel<false, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
  } else {
    if (vec) hipLaunchKernelGGL((act_rows_kernel<true, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((act_rows_kernel<false, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
  }
  RFX_CHECK_LAUNCH();
  return 0;
}
extern "C" int rfx_act_rows(const float* x, int64_t xs0, int64_t xs1, int64_t xs2, const float* gy, int64_t gs0,
                            int64_t gs1, int64_t gs2, float* out, int64_t os0, int64_t os1, int64_t os2, int32_t D0,
                            int32_t D1, int32_t D2, int32_t T, int32_t act, void* stream) {
  return act_rows_launch(false, x, xs0, xs1, xs2, gy, gs0, gs1, gs2, out, os0, os1, os2, D0, D1, D2, T, act, stream);
}
// x stored as bf16; gy == NULL: out (fp32) = act(x); otherwise out (bf16, strides in bf16 elements) = gy * act'(x)
extern "C" int rfx_act_rows16(const void* x, int64_t xs0, int64_t xs1, int64_t xs2, const float* gy, int64_t gs0,
                              int64_t gs1, int64_t gs2, void* out, int64_t os0, int64_t os1, int64_t os2, int32_t D0,
                              int32_t D1, int32_t D2, int32_t T, int32_t act, void* stream) {
  return act_rows_launch(true, x, xs0, xs1, xs2, gy, gs0, gs1, gs2, out, os0, os1, os2, D0, D1, D2, T, act, stream);
}
extern "C" int rfx_act_bwd(const float* x, const float* gy, float* gx, int64_t n, int32_t act, void* stream) {
  if (!x || !gy || !gx || n < 0) return -1;
  if (n == 0) return 0;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, gy, gx, n, act);
  RFX_CHECK_LAUNCH();
  return 0;
}
extern "C" int rfx_act_add_fwd(const float* x, const float* res, float* y, int64_t n, int32_t act, void* stream) {
  if (!x || !res || !y || n < 0) return -1;
  if (n == 0) return 0;
  hipLaunchKernelGGL(act_add_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, res, y, n, act);
  RFX_CHECK_LAUNCH();
  return 0;
}
__global__ __launch_bounds__(256) void mul_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, int64_t n) {
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
    if (i + 3 < n) {
      const f32x4 u = rfx_ld4(a + i), v = rfx_ld4(b + i);
      rfx_st4(y + i, u * v);
    } else {
      for (int64_t k = i; k < n; ++k) y[k] = a[k] * b[k];
    }
  }
}
extern "C" int rfx_mul(const float* a, const float* b, float* y, int64_t n, void* stream) {
  if (!a || !b || !y || n < 0) return -1;
  if (n == 0) return 0;
  int64_t g = (n + 1023) / 1024;
  hipLaunchKernelGGL(mul_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, (hipStream_t)stream, a, b, y, n);
  RFX_CHECK_LAUNCH();
  return 0;
}
__global__ void prelu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ slope, float* __restrict__ y, int64_t C, int64_t L) {
  const int64_t row = blockIdx.x;
  const float s = slope[row % C];
  const float* xr = x + row * L;
  float* yr = y + row * L;
  for (int64_t i = threadIdx.x; i < L; i += blockDim.x) {
    const float v = xr[i];
    yr[i] = v >= 0.f ? v : s * v;
  }
}
extern "C" int rfx_prelu_fwd(const float* x, const float* slope, float* y, int64_t N, int64_t C, int64_t L, void* stream) {
  if (!x || !slope || !y || N <= 0 || C <= 0 || L <= 0 || N * C > 0x7fffffffLL) return -1;
  hipLaunchKernelGGL(prelu_fwd_kernel, dim3((unsigned)(N * C)), dim3(256), 0, (hipStream_t)stream, x, slope, y, C, L);
  RFX_CHECK_LAUNCH();
  return 0;
}
extern "C" int rfx_prelu_bwd(const float* x, const float* gy, const float* slope, float* gx,
                             double* ws /* N * C doubles */, float* gslope, int64_t N, int64_t C, int64_t L, void* stream) {
  if (!x || !gy || !slope || !gx || !ws || !gslope || N <= 0 || C <= 0 || L <= 0 || N * C > 0x7fffffff) return -1;
  hipLaunchKernelGGL(prelu_bwd_kernel, dim3((unsigned)(N * C)), dim3(256), 0, (hipStream_t)stream, x, gy,
                     slope, gx, ws, N, C, L);
  RFX_CHECK_LAUNCH();
  hipLaunchKernelGGL(rfx_slot_sum_kernel<float>, RFX_SLOT_SUM_GRID((int)C), 0, (hipStream_t)stream, ws, (int)C, (int)N, 1, gslope);
  RFX_CHECK_LAUNCH();
  return 0;
}
// out[0] = scale * sum |a - b|; ws: RFX_L1_SLOTS doubles (per-workgroup partials, no initialisation needed)
extern "C" int rfx_l1_sum(const float* a, const float* b, int64_t n, double* ws, float scale, float* out, void* stream) {
  if (!a || !b || !out || !ws || n < 0) return -1;
  if (n == 0) return 0;
  int g = grid_for(n);
  g = g > RFX_L1_SLOTS ? RFX_L1_SLOTS : g;
  hipLaunchKernelGGL(l1_sum_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, a, b, n, ws);
  RFX_CHECK_LAUNCH();
  hipLaunchKernelGGL(l1_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ws, g, scale, out);
  RFX_CHECK_LAUNCH();
  return 0;
}

// contiguous (a, b) planes: grid = (C, N * nseg); a workgroup sums one 16-byte-vectorised segment of one (n, c) plane in fp32
// lanes (<= 4096 values per lane) and goes to fp64 for the cross-lane / cross-workgroup part.  The strided kernel above walks single
// dwords and measured 1.9 TB/s on the transposed-conv bias gradients (r02 profile).
__global__ __launch_bounds__(256) void channel_sum_plane_kernel(const float* __restrict__ x, int nseg, int64_t per, int64_t ns,
                                                                int64_t cs, double* __restrict__ slots) {
  const int c = blockIdx.x, n = blockIdx.y / nseg, sg = blockIdx.y - n * nseg;
  const int64_t per4 = per >> 2, seg4 = (per4 + nseg - 1) / nseg;
  const int64_t q0 = (int64_t)sg * seg4, q1 = q0 + seg4 < per4 ? q0 + seg4 : per4;
  const float* xp = x + (int64_t)n * ns + (int64_t)c * cs;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(xp);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int64_t q = q0 + threadIdx.x;
  for (; q + 768 < q1; q += 1024) {                        // 4 independent 16-byte loads in flight per lane
    const f32x4 v0 = x4[q], v1 = x4[q + 256], v2 = x4[q + 512], v3 = x4[q + 768];
    a0 += (v0[0] + v0[1]) + (v0[2] + v0[3]);
    a1 += (v1[0] + v1[1]) + (v1[2] + v1[3]);
    a2 += (v2[0] + v2[1]) + (v2[2] + v2[3]);
    a3 += (v3[0] + v3[1]) + (v3[2] + v3[3]);
  }
  for (; q < q1; q += 256) { const f32x4 v = x4[q]; a0 += (v[0] + v[1]) + (v[2] + v[3]); }
  if (sg == nseg - 1)                                       // the <= 3 values past the last whole vector
    for (int64_t t = (per4 << 2) + threadIdx.x; t < per; t += 256) a1 += xp[t];
  const double v[1] = {((double)a0 + (double)a1) + ((double)a2 + (double)a3)};
  rfx_block_store_slot<1>(v, slots, c, gridDim.y, blockIdx.y);
}

// geometry of rfx_channel_sum: workgroups (= slots) per channel; nseg != 0: the vectorised contiguous-plane kernel
static int channel_sum_geometry(const float* x, int32_t N, int32_t Cn, int32_t A, int32_t B, int64_t ns, int64_t cs, int64_t as, int64_t bs,
                                int64_t* nseg_out) {
  const int64_t total = (int64_t)N * A * B, per = (int64_t)A * B;
  *nseg_out = 0;
  if (bs == 1 && as == (int64_t)B && per >= 1024 && ns % 4 == 0 && cs % 4 == 0 && ((uintptr_t)x & 15) == 0) {
    int64_t nseg = per / (4 * 256 * 16);                    // ~16 vectors per lane and segment ...
    const int64_t want = (2048 + (int64_t)N * Cn - 1) / ((int64_t)N * Cn);     // ... but at least ~2048 workgroups in all
    nseg = nseg < want ? want : nseg;
    nseg = nseg < 1 ? 1 : (nseg > per / 1024 ? per / 1024 : nseg);
    if ((int64_t)N * nseg <= 65535) { *nseg_out = nseg; return (int)(N * nseg); }
  }
  int chunks = (int)(total / 16384);
  return chunks < 1 ? 1 : (chunks > 64 ? 64 : chunks);
}
extern "C" int64_t rfx_channel_sum_ws(const float* x, int32_t N, int32_t Cn, int32_t A, int32_t B, int64_t ns, int64_t cs, int64_t as,
                                      int64_t bs) {
  if (N <= 0 || Cn <= 0 || A <= 0 || B <= 0) return -1;
  int64_t nseg;
  return (int64_t)Cn * channel_sum_geometry(x, N, Cn, A, B, ns, cs, as, bs, &nseg);
}
extern "C" int rfx_channel_sum(const float* x, int32_t N, int32_t Cn, int32_t A, int32_t B, int64_t ns,
                               int64_t cs, int64_t as, int64_t bs, double* ws, float* out, void* stream) {
  if (!x || !out || !ws || N <= 0 || Cn <= 0 || A <= 0 || B <= 0) return -1;
  int64_t nseg;
  const int nslots = channel_sum_geometry(x, N, Cn, A, B, ns, cs, as, bs, &nseg);
  if (nseg) hipLaunchKernelGGL(channel_sum_plane_kernel, dim3(Cn, (unsigned)nslots), dim3(256), 0, (hipStream_t)stream, x, (int)nseg,
                               (int64_t)A * B, ns, cs, ws);
  else hipLaunchKernelGGL(channel_sum_kernel, dim3(Cn, nslots), dim3(256), 0, (hipStream_t)stream, x, N, Cn, A, B, ns, cs, as, bs, ws);
  RFX_CHECK_LAUNCH();
  hipLaunchKernelGGL(rfx_slot_sum_kernel<float>, RFX_SLOT_SUM_GRID(Cn), 0, (hipStream_t)stream, ws, Cn, nslots, 1, out);
  RFX_CHECK_LAUNCH();
  return 0;
}

// same-shape operand: flat 16-byte vectors
__global__ void add_flat_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out,
                                int64_t n, float alpha) {
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    const f32x4 w = reinterpret_cast<const f32x4*>(y)[i];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] += alpha * w[c];
    reinterpret_cast<f32x4*>(out)[i] = v;
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = x[i] + alpha * y[i];
}
// broadcast operand: one wave per ((n, c, a) row, 256-sample chunk of b); the row decode is wave-uniform 32-bit
// arithmetic instead of three 64-bit divisions per element
__global__ __launch_bounds__(256) void add_bcast_rows_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                             float* __restrict__ out, uint32_t nitems, uint32_t ipr,
                                                             int C, int A, int B, int64_t yn, int64_t yc, int64_t ya,
                                                             int64_t yb, float alpha) {
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t w = blockIdx.x * 4u + wave;
  if (w >= nitems) return;
  const uint32_t row = w / ipr, ck = w - row * ipr;
  const uint32_t r1 = row / (uint32_t)A, a = row - r1 * (uint32_t)A;
  const uint32_t n = r1 / (uint32_t)C, c = r1 - n * (uint32_t)C;
  const float* xr = x + (int64_t)row * B;
  float* orow = out + (int64_t)row * B;
  const float* yr = y + (int64_t)n * yn + (int64_t)c * yc + (int64_t)a * ya;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int b = (int)ck * 256 + q * 64 + (int)(threadIdx.x & 63);
    if (b < B) orow[b] = xr[b] + alpha * yr[(int64_t)b * yb];
  }
}
extern "C" int rfx_add_bcast(const float* x, const float* y, float* out, int64_t N, int32_t Cn, int32_t A, int32_t B,
                             int64_t yn, int64_t yc, int64_t ya, int64_t yb, float alpha, void* stream) {
  if (!x || !y || !out || N <= 0 || Cn <= 0 || A <= 0 || B <= 0) return -1;
  const int64_t total = N * Cn * A * B;
  const int64_t rows = N * Cn * A, per = ((int64_t)B + 255) / 256;
  if (yb == 1 && ya == B && yc == (int64_t)A * B && yn == (int64_t)Cn * A * B)
    hipLaunchKernelGGL(add_flat_kernel, dim3(grid_for(total / 4 + 1)), dim3(256), 0, (hipStream_t)stream, x, y, out, total,
                       alpha);
  else if (rows * per <= 0x7fffffffLL)
    hipLaunchKernelGGL(add_bcast_rows_kernel, dim3((unsigned)((rows * per + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x,
                       y, out, (uint32_t)(rows * per), (uint32_t)per, Cn, A, B, yn, yc, ya, yb, alpha);
  else
    hipLaunchKernelGGL(add_bcast_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, out, total, Cn, A,
                       B, yn, yc, ya, yb, alpha);
  RFX_CHECK_LAUNCH();
  return 0;
}
__global__ void span_mask_kernel(float* __restrict__ x, int F, int T, const int32_t* __restrict__ f0,
                                 const int32_t* __restrict__ f1, const int32_t* __restrict__ t0,
                                 const int32_t* __restrict__ t1) {
  const int r = blockIdx.y, f = blockIdx.x;
  const bool frow = f >= f0[r] && f < f1[r];
  int ta = frow ? 0 : t0[r], tb = frow ? T : t1[r];
  ta = ta < 0 ? 0 : ta; tb = tb > T ? T : tb;             // spans are clamped to the row
  float* row = x + ((int64_t)r * F + f) * T;
  for (int t = ta + threadIdx.x; t < tb; t += blockDim.x) row[t] = 0.f;
}
extern "C" int rfx_span_mask(float* x, int32_t R, int32_t F, int32_t T, const int32_t* f0, const int32_t* f1,
                             const int32_t* t0, const int32_t* t1, void* stream) {
  if (!x || !f0 || !f1 || !t0 || !t1 || R <= 0 || F <= 0 || T <= 0) return -1;
  hipLaunchKernelGGL(span_mask_kernel, dim3(F, R), dim3(256), 0, (hipStream_t)stream, x, F, T, f0, f1, t0, t1);
  RFX_CHECK_LAUNCH();
  return 0;
}
// frame k of row b covers tau in [k*stride, k*stride + width); the stitched output takes tau from frame
// k = clamp((tau - lim) / stride, 0, nfr - 1), lim = stride / 2 (first frame up to width - lim, last frame from lim on)
__global__ void blstm_frames_kernel(const float* __restrict__ src, const float* __restrict__ skip, float* __restrict__ dst,
                                    int B, int Cn, int T, int nfr, int width, int stride, int mode) {
  const int Bn = B * nfr, lim = stride / 2;
  const int64_t total = (mode == 0 || mode == 3) ? (int64_t)Cn * width * Bn : (int64_t)B * Cn * T;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    if (mode == 0 || mode == 3) {                      // destination h[c][t * Bn + b * nfr + k]
      const int bk = (int)(i % Bn);
      const int64_t r = i / Bn;
      const int t = (int)(r % width), c = (int)(r / width);
      const int b = bk / nfr, k = bk - b * nfr, tau = k * stride + t;
      float v = 0.f;
      if (tau < T) {
        if (mode == 0) v = src[((int64_t)b * Cn + c) * T + tau];
        else {
          int ks = (tau - lim) / stride;
          ks = tau < lim ? 0 : (ks > nfr - 1 ? nfr - 1 : ks);
          if (ks == k) v = src[((int64_t)b * Cn + c) * T + tau];
        }
      }
      dst[i] = v;
    } else {                                           // destination x[b][c][tau]
      const int tau = (int)(i % T);
      const int64_t r = i / T;
      const int c = (int)(r % Cn), b = (int)(r / Cn);
      float v = 0.f;
      if (mode == 2) {
        int k = (tau - lim) / stride;
        k = tau < lim ? 0 : (k > nfr - 1 ? nfr - 1 : k);
        v = src[(int64_t)c * width * Bn + (int64_t)(tau - k * stride) * Bn + b * nfr + k];
        if (skip) v += skip[i];
      } else {
        for (int k = 0; k < nfr; ++k) {
          const int t = tau - k * stride;
          if (t >= 0 && t < width) v += src[(int64_t)c * width * Bn + (int64_t)t * Bn + b * nfr + k];
        }
      }
      dst[i] = v;
    }
  }
}
extern "C" int rfx_blstm_frames(const float* src, const float* skip, float* dst, int32_t B, int32_t Cn, int32_t T, int32_t nfr,
                                int32_t width, int32_t stride, int32_t mode, void* stream) {
  if (!src || !dst || B <= 0 || Cn <= 0 || T <= 0 || nfr <= 0 || width <= 0 || stride <= 0 || mode < 0 || mode > 3) return -1;
  const int64_t total = (mode == 0 || mode == 3) ? (int64_t)Cn * width * B * nfr : (int64_t)B * Cn * T;
  hipLaunchKernelGGL(blstm_frames_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, skip, dst, B, Cn, T,
                     nfr, width, stride, mode);
  RFX_CHECK_LAUNCH();
  return 0;
}
extern "C" int rfx_row_moments_slots(int64_t L) {
  const int64_t gx = (L + 16383) / 16384;
  return (int)(gx < 1 ? 1 : (gx > 256 ? 256 : gx));
}
extern "C" int rfx_row_moments(const float* x, int32_t R, int64_t L, double* sums, float* mean, float* stdv, float eps, float* coef_a,
                               float* coef_b, void* stream) {
  if (!x || !sums || !mean || !stdv || R <= 0 || L <= 1 || (!coef_a) != (!coef_b)) return -1;
  hipStream_t s = (hipStream_t)stream;
  const int gx = rfx_row_moments_slots(L);
  hipLaunchKernelGGL(row_moments_kernel, dim3(gx, R), dim3(256), 0, s, x, L, sums);
  RFX_CHECK_LAUNCH();
  hipLaunchKernelGGL(row_moments_finalize_kernel, dim3((R + 3) / 4), dim3(256), 0, s, sums, R, gx, (double)L, mean, stdv, eps, coef_a,
                     coef_b);
  RFX_CHECK_LAUNCH();
  return 0;
}
extern "C" int rfx_row_affine(const float* x, const float* a, const float* b, float* out, int32_t R, int64_t L,
                              void* stream) {
  if (!x || !a || !out || R <= 0 || L <= 0) return -1;
  hipLaunchKernelGGL(row_affine_kernel, dim3(grid_for((int64_t)R * L)), dim3(256), 0, (hipStream_t)stream, x, a, b, out,
                     L, (int64_t)R * L);
  RFX_CHECK_LAUNCH();
  return 0;
}

// Inverted dropout, out[i] = keep(i) ? x[i] / (1 - p) : 0, with a COUNTER-BASED mask: keep(i) = u(seed, i) >= p where u is
// the top 24 bits of splitmix64(seed + i).  No mask tensor: the backward pass is the same launch on the gradient with the
// same seed.  (nn.LSTM inter-layer dropout of Open-Unmix, F.dropout in Cnn14 with train=True: classifier.py:211-284 /
// the un-vendored open-unmix; the reference draws from torch's Philox stream -- same distribution, different draws.)
__device__ __forceinline__ float dropout_u(uint64_t seed, uint64_t i) {
  uint64_t z = seed + 0x9e3779b97f4a7c15ull * (i + 1);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  z ^= z >> 31;
  return (float)(uint32_t)(z >> 40) * (1.0f / 16777216.0f);
}
__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n, float p, float inv_keep,
                               uint64_t seed) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = dropout_u(seed, (uint64_t)i) >= p ? x[i] * inv_keep : 0.f;
}
extern "C" int rfx_dropout(const float* x, float* out, int64_t n, float p, uint64_t seed, void* stream) {
  if (!x || !out || n < 0 || !(p >= 0.f) || !(p < 1.f)) return -1;
  if (n == 0) return 0;
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, x, out, n, p,
                     1.0f / (1.0f - p), seed);
  RFX_CHECK_LAUNCH();
  return 0;
}
