// Complex-valued pieces of DCUNet (asteroid.models.DCUNet via remfx/models.py:347-367).
// A complex tensor is stored as a real one with the channel axis doubled: (N, 2C, S),
// channels [0, C) = real parts, [C, 2C) = imaginary parts.  With that layout a complex
// convolution IS one real gather-GEMM with the block weight [[Wr, -Wi], [Wi, Wr]], so only
// the norm / activation / mask pieces need their own kernels:
//   * raw second-order moments per complex channel (ComplexBatchNorm statistics),
//   * y = leaky_relu(Z x + b) with a per-channel 2x2 real matrix Z (the whitening x affine
//     product, formed on C-length vectors by the host) -- forward and backward,
//   * the bounded mask  tanh(|m|) m/|m|  applied to the mixture STFT -- forward and backward.
// All HBM-bound streaming kernels: lanes along the contiguous spatial axis.
#include "common.h"

constexpr int CX_CHUNK = 4096;

// slots[(c * N * nchunks + n * nchunks + chunk) * 5 + {0..4}] = { sum xr, sum xi, sum xr^2, sum xr*xi, sum xi^2 } of one (n, c) row
// chunk (fp64; the wave's own slot -- a plain store, added in slot order by rfx_slot_sum_kernel: DESIGN.md 4.11)
__global__ __launch_bounds__(256) void cplx_moments_kernel(const float* __restrict__ x, int N, int C, int64_t S,
                                                           int nchunks, double* __restrict__ slots) {
  const int lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= (int64_t)N * C * nchunks) return;
  const int sc = (int)(item % nchunks);
  const int64_t r = item / nchunks;
  const int c = (int)(r % C), n = (int)(r / C);
  const float* xr = x + ((int64_t)n * 2 * C + c) * S;
  const float* xi = xr + (int64_t)C * S;
  const int64_t s0 = (int64_t)sc * CX_CHUNK, s1 = min(s0 + CX_CHUNK, S);
  float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int64_t s = s0 + lane; s < s1; s += 64) {
    const float a = xr[s], b = xi[s];
    v[0] += a; v[1] += b; v[2] += a * a; v[3] += a * b; v[4] += b * b;
  }
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    const double d = rfx_wave_sum_d((double)v[q]);
    if (lane == 0) slots[(((int64_t)c * N + n) * nchunks + sc) * 5 + q] = d;
  }
}

// gx += d(sum_q coef_q * moment_q)/dx :  gxr = c0 + 2 c2 xr + c3 xi ; gxi = c1 + 2 c4 xi + c3 xr
__global__ __launch_bounds__(256) void cplx_moments_bwd_kernel(const float* __restrict__ x, const float* __restrict__ coef,
                                                               int N, int C, int64_t S, int nchunks,
                                                               float* __restrict__ gx) {
  // one workgroup = one (n, c) row chunk (no per-element index decoding, coefficients wave-uniform)
  const int sc = blockIdx.x % nchunks;
  const int r = blockIdx.x / nchunks;
  const int c = r % C, n = r / C;
  const float k0 = coef[c], k1 = coef[C + c], k2 = coef[2 * C + c], k3 = coef[3 * C + c], k4 = coef[4 * C + c];
  const int64_t base = ((int64_t)n * 2 * C + c) * S, im = (int64_t)C * S;
  const int64_t s0 = (int64_t)sc * CX_CHUNK, s1 = s0 + CX_CHUNK < S ? s0 + CX_CHUNK : S;
  for (int64_t s = s0 + threadIdx.x; s < s1; s += 256) {
    const float a = x[base + s], b = x[base + im + s];
    gx[base + s] += k0 + 2.f * k2 * a + k3 * b;
    gx[base + im + s] += k1 + 2.f * k4 * b + k3 * a;
  }
}

// coef: (6, C) = Zrr, Zri, Zir, Zii, Br, Bi.  out may be a channel slice of a larger buffer:
// out[n*out_ns + ch*S + s], ch in [0, 2C) with the imaginary half at out_im_off channels.
// One workgroup = one (n, c) row chunk: the six coefficients are wave-uniform and there is no index arithmetic per
// element (the first version decoded (n, c, s) from a flat index with two 64-bit divisions per element and ran
// VALU-bound: 0.7 ms per call in the chain-inference profile).
__global__ __launch_bounds__(256) void cplx_affine_act_kernel(const float* __restrict__ x, const float* __restrict__ coef,
                                                              int N, int C, int64_t S, int nchunks, float slope,
                                                              float* __restrict__ out, int64_t out_ns, int64_t out_im_off) {
  const int sc = blockIdx.x % nchunks;
  const int r = blockIdx.x / nchunks;
  const int c = r % C, n = r / C;
  const float zrr = coef[c], zri = coef[C + c], zir = coef[2 * C + c], zii = coef[3 * C + c];
  const float br = coef[4 * C + c], bi = coef[5 * C + c];
  const float* xr = x + ((int64_t)n * 2 * C + c) * S;
  const float* xi = xr + (int64_t)C * S;
  float* orp = out + (int64_t)n * out_ns + (int64_t)c * S;
  float* oip = orp + out_im_off * S;
  const int64_t s0 = (int64_t)sc * CX_CHUNK, s1 = s0 + CX_CHUNK < S ? s0 + CX_CHUNK : S;
  for (int64_t s = s0 + threadIdx.x; s < s1; s += 256) {
    const float a = xr[s], b = xi[s];
    float ur = zrr * a + zri * b + br;
    float ui = zir * a + zii * b + bi;
    orp[s] = ur >= 0.f ? ur : slope * ur;
    oip[s] = ui >= 0.f ? ui : slope * ui;
  }
}

// backward of the above: gx (N, 2C, S) written; gcoef (6, C) accumulated with atomics (zeroed by caller)
__global__ __launch_bounds__(256) void cplx_affine_act_bwd_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ coef,
                                                                  const float* __restrict__ gy, int64_t gy_ns,
                                                                  int64_t gy_im_off, int N, int C, int64_t S,
                                                                  int nchunks, float slope, float* __restrict__ gx,
                                                                  double* __restrict__ slots /* [6][C][N * nchunks] */) {
  const int lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= (int64_t)N * C * nchunks) return;
  const int sc = (int)(item % nchunks);
  const int64_t r = item / nchunks;
  const int c = (int)(r % C), n = (int)(r / C);
  const float zrr = coef[c], zri = coef[C + c], zir = coef[2 * C + c], zii = coef[3 * C + c];
  const float br = coef[4 * C + c], bi = coef[5 * C + c];
  const int64_t base = ((int64_t)n * 2 * C + c) * S, gbase = (int64_t)n * gy_ns + (int64_t)c * S;
  const int64_t s0 = (int64_t)sc * CX_CHUNK, s1 = min(s0 + CX_CHUNK, S);
  float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int64_t s = s0 + lane; s < s1; s += 64) {
    const float a = x[base + s], b = x[base + (int64_t)C * S + s];
    const float ur = zrr * a + zri * b + br, ui = zir * a + zii * b + bi;
    const float gr = gy[gbase + s] * (ur >= 0.f ? 1.f : slope);
    const float gi = gy[gbase + gy_im_off * S + s] * (ui >= 0.f ? 1.f : slope);
    gx[base + s] = zrr * gr + zir * gi;
    gx[base + (int64_t)C * S + s] = zri * gr + zii * gi;
    v[0] += gr * a; v[1] += gr * b; v[2] += gi * a; v[3] += gi * b; v[4] += gr; v[5] += gi;
  }
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const float d = rfx_wave_sum(v[q]);
    if (lane == 0) slots[(((int64_t)q * C + c) * N + n) * nchunks + sc] = (double)d;
  }
}

// m, tf, out: (N, 2, P) planes (sample stride given).  out = tanh(|m|)/|m| * m (*) tf   (complex product)
__global__ void bound_mask_kernel(const float* __restrict__ m, const float* __restrict__ tf, float* __restrict__ out,
                                  int N, int64_t P, int64_t m_ns, int64_t tf_ns, int64_t out_ns) {
  const int64_t total = (int64_t)N * P;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t p = i % P, n = i / P;
    const float mr = m[n * m_ns + p], mi = m[n * m_ns + P + p];
    const float tr = tf[n * tf_ns + p], ti = tf[n * tf_ns + P + p];
    const float mag = sqrtf(mr * mr + mi * mi);
    const float k = tanhf(mag) / mag;
    const float ar = k * mr, ai = k * mi;
    out[n * out_ns + p] = ar * tr - ai * ti;
    out[n * out_ns + P + p] = ar * ti + ai * tr;
  }
}
__global__ void bound_mask_bwd_kernel(const float* __restrict__ m, const float* __restrict__ tf,
                                      const float* __restrict__ gout, float* __restrict__ gm, int N, int64_t P,
                                      int64_t m_ns, int64_t tf_ns, int64_t g_ns, int64_t gm_ns) {
  const int64_t total = (int64_t)N * P;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t p = i % P, n = i / P;
    const float mr = m[n * m_ns + p], mi = m[n * m_ns + P + p];
    const float tr = tf[n * tf_ns + p], ti = tf[n * tf_ns + P + p];
    const float gr = gout[n * g_ns + p], gi = gout[n * g_ns + P + p];
    const float gar = gr * tr + gi * ti, gai = -gr * ti + gi * tr;   // gout (*) conj(tf)
    const float mag = sqrtf(mr * mr + mi * mi);
    const float th = tanhf(mag);
    const float k = th / mag;
    const float kp = ((1.f - th * th) * mag - th) / (mag * mag);      // dk/dmag
    const float dot = (gar * mr + gai * mi) * kp / mag;
    gm[n * gm_ns + p] = gar * k + dot * mr;
    gm[n * gm_ns + P + p] = gai * k + dot * mi;
  }
}

// Open-Unmix Separator, niter = 0 "wiener": estimate = magnitude x phase of the mixture STFT
// (models.py:298 via umx Separator).  xc, out: complex [n] (view_as_real layout); mag: [n].
__global__ void phase_mask_kernel(const float* __restrict__ mag, const float2* __restrict__ xc,
                                  float2* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float2 x = xc[i];
    const float a = sqrtf(x.x * x.x + x.y * x.y);
    const float c = a > 0.f ? x.x / a : 1.f, s = a > 0.f ? x.y / a : 0.f;   // cos / sin of atan2(im, re)
    const float m = mag[i];
    out[i] = make_float2(m * c, m * s);
  }
}
__global__ void phase_mask_bwd_kernel(const float2* __restrict__ xc, const float2* __restrict__ gout,
                                      float* __restrict__ gmag, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float2 x = xc[i], g = gout[i];
    const float a = sqrtf(x.x * x.x + x.y * x.y);
    const float c = a > 0.f ? x.x / a : 1.f, s = a > 0.f ? x.y / a : 0.f;
    gmag[i] = g.x * c + g.y * s;
  }
}

static int cx_grid(int64_t total) {
  const int64_t b = (total + 1023) / 1024;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// slots per channel of the two reductions below: one per (sample, 4096-value chunk)
extern "C" int64_t rfx_cplx_slots(int32_t N, int64_t S) {
  if (N <= 0 || S <= 0) return -1;
  return (int64_t)N * ((S + CX_CHUNK - 1) / CX_CHUNK);
}
extern "C" int rfx_cplx_moments(const float* x, int32_t N, int32_t C, int64_t S, double* ws, double* sums, void* stream) {
  if (!x || !ws || !sums || N <= 0 || C <= 0 || S <= 0) return -1;
  hipStream_t s = (hipStream_t)stream;
  const int nchunks = (int)((S + CX_CHUNK - 1) / CX_CHUNK);
  const int64_t items = (int64_t)N * C * nchunks;
  if ((int64_t)N * nchunks > 0x7fffffff) return -1;
  hipLaunchKernelGGL(cplx_moments_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s, x, N, C, S, nchunks, ws);
  RFX_CHECK_LAUNCH();
  hipLaunchKernelGGL(rfx_slot_sum_kernel<double>, RFX_SLOT_SUM_GRID(5 * C), 0, s, ws, C, N * nchunks, 5, sums);
  RFX_CHECK_LAUNCH();
  return 0;
}
extern "C" int rfx_cplx_moments_bwd(const float* x, const float* coef, int32_t N, int32_t C, int64_t S, float* gx,
                                    void* stream) {
  if (!x || !coef || !gx || N <= 0 || C <= 0 || S <= 0) return -1;
  const int nchunks = (int)((S + CX_CHUNK - 1) / CX_CHUNK);
  if ((int64_t)N * C * nchunks > 0x7fffffff) return -1;
  hipLaunchKernelGGL(cplx_moments_bwd_kernel, dim3((unsigned)((int64_t)N * C * nchunks)), dim3(256), 0, (hipStream_t)stream, x,
                     coef, N, C, S, nchunks, gx);
  RFX_CHECK_LAUNCH();
  return 0;
}
extern "C" int rfx_cplx_affine_act_fwd(const float* x, const float* coef, int32_t N, int32_t C, int64_t S,
                                       float slope, float* out, int64_t out_ns, int64_t out_im_off, void* stream) {
  if (!x || !coef || !out || N <= 0 || C <= 0 || S <= 0) return -1;
  const int nchunks = (int)((S + CX_CHUNK - 1) / CX_CHUNK);
  const int64_t blocks = (int64_t)N * C * nchunks;
  if (blocks > 0x7fffffff) return -1;
  hipLaunchKernelGGL(cplx_affine_act_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, coef, N, C, S,
                     nchunks, slope, out, out_ns, out_im_off);
  RFX_CHECK_LAUNCH();
  return 0;
}
extern "C" int rfx_cplx_affine_act_bwd(const float* x, const float* coef, const float* gy, int64_t gy_ns,
                                       int64_t gy_im_off, int32_t N, int32_t C, int64_t S, float slope, float* gx,
                                       double* ws, float* gcoef, void* stream) {
  if (!x || !coef || !gy || !gx || !ws || !gcoef || N <= 0 || C <= 0 || S <= 0) return -1;
  hipStream_t s = (hipStream_t)stream;
  const int nchunks = (int)((S + CX_CHUNK - 1) / CX_CHUNK);
  const int64_t items = (int64_t)N * C * nchunks;
  if ((int64_t)N * nchunks > 0x7fffffff) return -1;
  hipLaunchKernelGGL(cplx_affine_act_bwd_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s, x, coef, gy, gy_ns,
                     gy_im_off, N, C, S, nchunks, slope, gx, ws);
  RFX_CHECK_LAUNCH();
  hipLaunchKernelGGL(rfx_slot_sum_kernel<float>, RFX_SLOT_SUM_GRID(6 * C), 0, s, ws, 6 * C, N * nchunks, 1, gcoef);
  RFX_CHECK_LAUNCH();
  return 0;
}
extern "C" int rfx_bound_mask_fwd(const float* m, const float* tf, float* out, int32_t N, int64_t P, int64_t m_ns,
                                  int64_t tf_ns, int64_t out_ns, void* stream) {
  if (!m || !tf || !out || N <= 0 || P <= 0) return -1;
  hipLaunchKernelGGL(bound_mask_kernel, dim3(cx_grid((int64_t)N * P)), dim3(256), 0, (hipStream_t)stream, m, tf, out, N,
                     P, m_ns, tf_ns, out_ns);
  RFX_CHECK_LAUNCH();
  return 0;
}
extern "C" int rfx_bound_mask_bwd(const float* m, const float* tf, const float* gout, float* gm, int32_t N, int64_t P,
                                  int64_t m_ns, int64_t tf_ns, int64_t g_ns, int64_t gm_ns, void* stream) {
  if (!m || !tf || !gout || !gm || N <= 0 || P <= 0) return -1;
  hipLaunchKernelGGL(bound_mask_bwd_kernel, dim3(cx_grid((int64_t)N * P)), dim3(256), 0, (hipStream_t)stream, m, tf,
                     gout, gm, N, P, m_ns, tf_ns, g_ns, gm_ns);
  RFX_CHECK_LAUNCH();
  return 0;
}

extern "C" int rfx_phase_mask_fwd(const float* mag, const float* xc, float* out, int64_t n, void* stream) {
  if (!mag || !xc || !out || n <= 0) return -1;
  hipLaunchKernelGGL(phase_mask_kernel, dim3(cx_grid(n)), dim3(256), 0, (hipStream_t)stream, mag, (const float2*)xc,
                     (float2*)out, n);
  RFX_CHECK_LAUNCH();
  return 0;
}
extern "C" int rfx_phase_mask_bwd(const float* xc, const float* gout, float* gmag, int64_t n, void* stream) {
  if (!xc || !gout || !gmag || n <= 0) return -1;
  hipLaunchKernelGGL(phase_mask_bwd_kernel, dim3(cx_grid(n)), dim3(256), 0, (hipStream_t)stream, (const float2*)xc,
                     (const float2*)gout, gmag, n);
  RFX_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------
// ComplexBatchNorm coefficients and their backward (asteroid DCUNet's complex_nn.BatchNorm via remfx/models.py:356-367):
// per channel, y = Z x + B' with Z = W . V^{-1/2} (2x2 inverse square root of the covariance) and the mean folded into B'.
// One thread per channel; until round 4 this was ~45 (forward) + ~60 (autograd) torch launches on (C,) tensors per norm layer --
// 4000 of the 4300 launches of a DCUNet step.
// sums: [c][5] fp64 = sum xr, xi, xr^2, xr xi, xi^2 (rfx_cplx_moments), inv = 1 / (N S); or stats_in (5, C) fp32 = running
// statistics (eval).  The variances are formed in fp64 (m2 - m^2 cancels), everything after in fp32 as the reference does.
// ---------------------------------------------------------------------------------
struct CxCoefW { const float *Wrr, *Wri, *Wii, *Br, *Bi; };
struct CxCoefRun { float *RMr, *RMi, *RVrr, *RVri, *RVii; };

__device__ __forceinline__ void cx_stats(const double* sums, double inv, const float* stats_in, int C, int c, float st[5]) {
  if (sums) {
    const double m0 = sums[c * 5 + 0] * inv, m1 = sums[c * 5 + 1] * inv;
    st[0] = (float)m0; st[1] = (float)m1;
    st[2] = (float)(sums[c * 5 + 2] * inv - m0 * m0);
    st[3] = (float)(sums[c * 5 + 3] * inv - m0 * m1);
    st[4] = (float)(sums[c * 5 + 4] * inv - m1 * m1);
  } else {
#pragma unroll
    for (int q = 0; q < 5; ++q) st[q] = stats_in[q * C + c];
  }
}

__global__ void cplx_coef_fwd_kernel(const double* __restrict__ sums, double inv, const float* __restrict__ stats_in, CxCoefW w,
                                     float eps, int C, float* __restrict__ coef, float* __restrict__ stats_out, CxCoefRun run,
                                     float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float st[5];
  cx_stats(sums, inv, stats_in, C, c, st);
  if (stats_out) {
#pragma unroll
    for (int q = 0; q < 5; ++q) stats_out[q * C + c] = st[q];
  }
  if (run.RMr) {                         // running statistics: buf.lerp_(stat, momentum)
    run.RMr[c] += momentum * (st[0] - run.RMr[c]);   run.RMi[c] += momentum * (st[1] - run.RMi[c]);
    run.RVrr[c] += momentum * (st[2] - run.RVrr[c]); run.RVri[c] += momentum * (st[3] - run.RVri[c]);
    run.RVii[c] += momentum * (st[4] - run.RVii[c]);
  }
  const float Mr = st[0], Mi = st[1], Vrr = st[2] + eps, Vri = st[3], Vii = st[4] + eps;
  const float tau = Vrr + Vii, delta = Vrr * Vii - Vri * Vri;
  const float s = sqrtf(delta), t = sqrtf(tau + 2.f * s), rst = 1.f / (s * t);
  const float Urr = (s + Vii) * rst, Uii = (s + Vrr) * rst, Uri = -Vri * rst;
  const float Wrr = w.Wrr[c], Wri = w.Wri[c], Wii = w.Wii[c];
  const float Zrr = Wrr * Urr + Wri * Uri, Zri = Wrr * Uri + Wri * Uii;
  const float Zir = Wri * Urr + Wii * Uri, Zii = Wri * Uri + Wii * Uii;
  coef[0 * C + c] = Zrr; coef[1 * C + c] = Zri; coef[2 * C + c] = Zir; coef[3 * C + c] = Zii;
  coef[4 * C + c] = w.Br[c] - (Zrr * Mr + Zri * Mi);
  coef[5 * C + c] = w.Bi[c] - (Zir * Mr + Zii * Mi);
}

// gcoef (6, C) -> gw (5, C) = d/d(Wrr, Wri, Wii, Br, Bi) and, in training mode, cm (5, C) = d/d(raw moment q) * inv (what
// rfx_cplx_moments_bwd takes).  Reverse-mode by hand of the forward above, in fp64.
__global__ void cplx_coef_bwd_kernel(const double* __restrict__ sums, double inv, const float* __restrict__ stats_in, CxCoefW w,
                                     float eps, int C, const float* __restrict__ gcoef, float* __restrict__ gw,
                                     float* __restrict__ cm) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float st[5];
  cx_stats(sums, inv, stats_in, C, c, st);
  const double Mr = st[0], Mi = st[1], Vrr = (double)(st[2] + eps), Vri = st[3], Vii = (double)(st[4] + eps);
  const double tau = Vrr + Vii, delta = Vrr * Vii - Vri * Vri;
  const double s = sqrt(delta), t = sqrt(tau + 2.0 * s), rst = 1.0 / (s * t);
  const double Urr = (s + Vii) * rst, Uii = (s + Vrr) * rst, Uri = -Vri * rst;
  const double Wrr = w.Wrr[c], Wri = w.Wri[c], Wii = w.Wii[c];
  const double Zrr = Wrr * Urr + Wri * Uri, Zri = Wrr * Uri + Wri * Uii;
  const double Zir = Wri * Urr + Wii * Uri, Zii = Wri * Uri + Wii * Uii;
  const double gB0 = gcoef[4 * C + c], gB1 = gcoef[5 * C + c];
  const double gZrr = (double)gcoef[0 * C + c] - gB0 * Mr, gZri = (double)gcoef[1 * C + c] - gB0 * Mi;
  const double gZir = (double)gcoef[2 * C + c] - gB1 * Mr, gZii = (double)gcoef[3 * C + c] - gB1 * Mi;
  gw[0 * C + c] = (float)(gZrr * Urr + gZri * Uri);
  gw[1 * C + c] = (float)(gZrr * Uri + gZri * Uii + gZir * Urr + gZii * Uri);
  gw[2 * C + c] = (float)(gZir * Uri + gZii * Uii);
  gw[3 * C + c] = (float)gB0;
  gw[4 * C + c] = (float)gB1;
  if (!cm) return;
  const double gMr = -(gB0 * Zrr + gB1 * Zir), gMi = -(gB0 * Zri + gB1 * Zii);
  const double gUrr = gZrr * Wrr + gZir * Wri;
  const double gUri = gZrr * Wri + gZri * Wrr + gZir * Wii + gZii * Wri;
  const double gUii = gZri * Wri + gZii * Wii;
  double gs = (gUrr + gUii) * rst;
  double gVii = gUrr * rst, gVrr = gUii * rst, gVri = -gUri * rst;
  const double grst = gUrr * (s + Vii) + gUii * (s + Vrr) - gUri * Vri;
  const double gst = -grst * rst * rst;                 // d / d(s t)
  gs += gst * t;
  const double gt = gst * s;
  const double gtau = gt / (2.0 * t);                   // d / d(tau + 2 s)
  gs += 2.0 * gtau;
  const double gdelta = gs / (2.0 * s);
  gVrr += gdelta * Vii + gtau;
  gVii += gdelta * Vrr + gtau;
  gVri += -2.0 * gdelta * Vri;
  // central statistics -> raw moments: Vrr = m2 - Mr^2, Vri = m3 - Mr Mi, Vii = m4 - Mi^2
  cm[0 * C + c] = (float)((gMr - 2.0 * Mr * gVrr - Mi * gVri) * inv);
  cm[1 * C + c] = (float)((gMi - 2.0 * Mi * gVii - Mr * gVri) * inv);
  cm[2 * C + c] = (float)(gVrr * inv);
  cm[3 * C + c] = (float)(gVri * inv);
  cm[4 * C + c] = (float)(gVii * inv);
}

extern "C" int rfx_cplx_coef_fwd(const double* sums, double inv_count, const float* stats_in, const float* Wrr, const float* Wri,
                                 const float* Wii, const float* Br, const float* Bi, float eps, int32_t C, float* coef,
                                 float* stats_out, float* RMr, float* RMi, float* RVrr, float* RVri, float* RVii, float momentum,
                                 void* stream) {
  if ((sums == nullptr) == (stats_in == nullptr) || !Wrr || !Wri || !Wii || !Br || !Bi || !coef || C <= 0) return -1;
  if ((RMr != nullptr) != (RMi != nullptr) || (RMr != nullptr) != (RVrr != nullptr) || (RMr != nullptr) != (RVri != nullptr) ||
      (RMr != nullptr) != (RVii != nullptr))
    return -1;
  CxCoefW w{Wrr, Wri, Wii, Br, Bi};
  CxCoefRun r{RMr, RMi, RVrr, RVri, RVii};
  hipLaunchKernelGGL(cplx_coef_fwd_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, sums, inv_count, stats_in, w, eps,
                     C, coef, stats_out, r, momentum);
  RFX_CHECK_LAUNCH();
  return 0;
}

extern "C" int rfx_cplx_coef_bwd(const double* sums, double inv_count, const float* stats_in, const float* Wrr, const float* Wri,
                                 const float* Wii, const float* Br, const float* Bi, float eps, int32_t C, const float* gcoef,
                                 float* gw, float* cm, void* stream) {
  if ((sums == nullptr) == (stats_in == nullptr) || !Wrr || !Wri || !Wii || !Br || !Bi || !gcoef || !gw || C <= 0) return -1;
  CxCoefW w{Wrr, Wri, Wii, Br, Bi};
  hipLaunchKernelGGL(cplx_coef_bwd_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, sums, inv_count, stats_in, w, eps,
                     C, gcoef, gw, cm);
  RFX_CHECK_LAUNCH();
  return 0;
}
