// Loss reductions of the hot path: auraloss STFTLoss terms (spectral convergence +
// log-magnitude L1), waveform L1, SI-SDR sums.  All HBM-bound streaming reductions.
// Replaces: models.py:107,299,320,362,385 (mrstft + 100*L1), models.py:227-255 (metrics).
#include "common.h"

// xc, yc: [R][n] complex (float2); sums[r] = { sum (ym-xm)^2, sum ym^2, sum |log xm - log ym| }
__global__ __launch_bounds__(256) void stft_loss_reduce_kernel(const float2* __restrict__ xc,
                                                               const float2* __restrict__ yc, int64_t n,
                                                               float eps, double* __restrict__ slots) {
  const int r = blockIdx.y;
  const float2* xr = xc + (int64_t)r * n;
  const float2* yr = yc + (int64_t)r * n;
  // |log xm - log ym| = |log px - log py| / 2 on the clamped powers; hardware sqrt / log (1 ulp) -- the kernel is
  // otherwise bound by the transcendental fix-up sequences, not by HBM.  Eight terms are summed in fp32, then in fp64.
  double a = 0, b = 0, c = 0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += 8 * stride) {
    float fa = 0.f, fb = 0.f, fc = 0.f;
    float2 xv[8], yv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {            // all loads first, unconditional (clamped index; masked below)
      const int64_t i = i0 + u * stride;
      xv[u] = xr[i < n ? i : n - 1];
      yv[u] = yr[i < n ? i : n - 1];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float m = (i0 + u * stride) < n ? 1.f : 0.f;
      const float2 x = xv[u], y = yv[u];
      const float px = fmaxf(x.x * x.x + x.y * x.y, eps), py = fmaxf(y.x * y.x + y.y * y.y, eps);
      const float dd = __builtin_amdgcn_sqrtf(py) - __builtin_amdgcn_sqrtf(px);
      fa += m * dd * dd;
      fb += m * py;
      fc += m * fabsf(__builtin_amdgcn_logf(px) - __builtin_amdgcn_logf(py));
    }
    a += (double)fa; b += (double)fb; c += (double)(fc * 0.34657359027997264f);   // log2 -> ln, halved
  }
  const double v[3] = {a, b, c};
  rfx_block_store_slot<3>(v, slots, r, gridDim.x, blockIdx.x);
}

// gxc = d/dxc [ w_sc * sqrt(A_r)/sqrt(B_r) + w_lm * sum |log xm - log ym| ]
//   w_sc, w_lm already contain the upstream gradient and the 1/R, 1/(R*n), 1/3 factors.
// ymag != NULL: the clamped target magnitudes sqrt(max(|Y|^2, eps)) as stored by rfx_stft_pair_loss (yc unused)
__global__ __launch_bounds__(256) void stft_loss_grad_kernel(const float2* __restrict__ xc,
                                                             const float2* __restrict__ yc, const float* __restrict__ ymag, int64_t n,
                                                             float eps, const float* __restrict__ sums,
                                                             float w_sc, float w_lm, const float* __restrict__ gup, float2* __restrict__ gxc) {
  const int r = blockIdx.y;
  if (gup) { const float u = gup[0]; w_sc *= u; w_lm *= u; }      // upstream scalar gradient read on the device (no host sync)
  const float A = sums[3 * r], B = sums[3 * r + 1];
  const float ksc = (A > 0.f && B > 0.f) ? w_sc / (sqrtf(A) * sqrtf(B)) : 0.f;   // d sqrt(A)/sqrt(B) / dA * 2
  const float2* xr = xc + (int64_t)r * n;
  const float2* yr = yc ? yc + (int64_t)r * n : nullptr;
  const float* mr = ymag ? ymag + (int64_t)r * n : nullptr;
  float2* gr = gxc + (int64_t)r * n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float2 x = xr[i];
    float2 g = make_float2(0.f, 0.f);
    if (mr) {
      // paired forward (rfx_stft_pair_loss): the same rounding sequence as its epilogue -- rfx_pow2, hardware sqrt -- and the sign of
      // log xm - log ym taken from xm - ym, so identical signals (|X| == |Y| bit for bit) get an exactly zero gradient
      const float px = rfx_pow2(x.x, x.y);
      if (px > eps) {
        const float xm = __builtin_amdgcn_sqrtf(px), ym = mr[i];
        const float dm = ksc * (xm - ym) + w_lm * (xm > ym ? 1.f : (xm < ym ? -1.f : 0.f)) / xm;
        g.x = dm * x.x / xm;
        g.y = dm * x.y / xm;
      }
    } else {
      const float px = x.x * x.x + x.y * x.y;
      if (px > eps) {   // clamp(min=eps) passes no gradient below eps
        const float xm = sqrtf(px);
        const float2 y = yr[i];
        const float ym = sqrtf(fmaxf(y.x * y.x + y.y * y.y, eps));
        const float lg = logf(xm) - logf(ym);
        const float dm = ksc * (xm - ym) + w_lm * (lg > 0.f ? 1.f : (lg < 0.f ? -1.f : 0.f)) / xm;
        g.x = dm * x.x / xm;
        g.y = dm * x.y / xm;
      }
    }
    gr[i] = g;
  }
}

// g[i] = w * sign(a[i]-b[i])
__global__ void l1_grad_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n, float w,
                               const float* __restrict__ gup, float* __restrict__ g) {
  if (gup) w *= gup[0];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float d = a[i] - b[i];
    g[i] = d > 0.f ? w : (d < 0.f ? -w : 0.f);
  }
}

// per row r of [R][L]: sums[r] = { sum x, sum t, sum x*t, sum x*x, sum t*t }  (double accumulate)
__global__ __launch_bounds__(256) void sisdr_sums_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                         int64_t L, int64_t xs, int64_t ts, double* __restrict__ sums /* slots */) {
  const int r = blockIdx.y;
  const float* xr = x + (int64_t)r * xs;
  const float* tr = t + (int64_t)r * ts;
  double s[5] = {0, 0, 0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < L; i += (int64_t)gridDim.x * 256) {
    const double a = xr[i], b = tr[i];
    s[0] += a; s[1] += b; s[2] += a * b; s[3] += a * a; s[4] += b * b;
  }
  rfx_block_store_slot<5>(s, sums, r, gridDim.x, blockIdx.x);
}

// The scalar tail of auraloss MultiResolutionSTFTLoss over the row sums of all resolutions in ONE launch (it was ~24 one-element torch
// launches per evaluation, three evaluations per training step): per resolution k, sums_k [R][3] = { sum (ym - xm)^2, sum ym^2,
// sum |log xm - log ym| } per row, n_k spectrum cells per row:
//   sc_k = per_example ? mean_r sqrt(A_r) / sqrt(B_r) : sqrt(sum_r A_r) / sqrt(sum_r B_r);   lm_k = sum_r C_r / (R n_k)
//   out[0] = (1 / nres) sum_k (sc_k + lm_k)
// One wave; rows are taken in order by lane l = r mod 64 and reduced with the fixed butterfly (bit-reproducible).
struct MrCombineArgs {
  const float* sums[8];
  double n[8];
  int nres, R, per_example;
  float* out;
};
__global__ __launch_bounds__(64) void mrstft_combine_kernel(const MrCombineArgs a) {
  const int lane = threadIdx.x;
  double total = 0.0;
  for (int k = 0; k < a.nres; ++k) {
    const float* s = a.sums[k];
    double sc = 0.0, sa = 0.0, sb = 0.0, sl = 0.0;
    for (int r = lane; r < a.R; r += 64) {
      const float A = s[3 * r], B = s[3 * r + 1], C = s[3 * r + 2];
      sc += (double)(sqrtf(A) / sqrtf(B));
      sa += (double)A; sb += (double)B; sl += (double)C;
    }
    sc = rfx_wave_sum_d(sc); sa = rfx_wave_sum_d(sa); sb = rfx_wave_sum_d(sb); sl = rfx_wave_sum_d(sl);
    const double scv = a.per_example ? sc / (double)a.R : (double)(sqrtf((float)sa) / sqrtf((float)sb));
    total += scv + sl / ((double)a.R * a.n[k]);
  }
  if (lane == 0) a.out[0] = (float)(total / (double)a.nres);
}

// The scalar tail of auraloss SISDRLoss over the fp64 row sums of rfx_sisdr_sums: out[0] = -mean_r 10 log10(|alpha t|^2 / (|x - alpha t|^2 + eps) + eps),
// alpha = <x, t> / (|t|^2 + eps), after removing the row means when zero_mean (the ~22 one-element torch launches it replaces ran
// twice per training step).  One wave, rows in lane order, fixed butterfly.
__global__ __launch_bounds__(64) void sisdr_finish_kernel(const double* __restrict__ sums, int R, double L, int zero_mean, double eps,
                                                          float* __restrict__ out) {
  const int lane = threadIdx.x;
  double acc = 0.0;
  for (int r = lane; r < R; r += 64) {
    const double sx = sums[5 * r], st = sums[5 * r + 1];
    double sxt = sums[5 * r + 2], sxx = sums[5 * r + 3], stt = sums[5 * r + 4];
    if (zero_mean) { sxt -= sx * st / L; sxx -= sx * sx / L; stt -= st * st / L; }
    const double alpha = sxt / (stt + eps);
    const double tt = alpha * alpha * stt;
    const double res = sxx - 2.0 * alpha * sxt + tt;
    acc += 10.0 * log10(tt / (res + eps) + eps);
  }
  acc = rfx_wave_sum_d(acc);
  if (lane == 0) out[0] = (float)(-acc / (double)R);
}

static int grid_x(int64_t n) {
  const int64_t b = (n + 2047) / 2048;
  return (int)(b < 1 ? 1 : (b > 512 ? 512 : b));
}
// reductions: fewer, longer workgroups per row (each ends in one slot store)
static int grid_red(int64_t n) {
  const int64_t b = (n + 16383) / 16384;
  return (int)(b < 1 ? 1 : (b > 64 ? 64 : b));
}

extern "C" int rfx_stft_loss_reduce(const float* xc, const float* yc, int32_t R, int64_t n, float eps, double* ws,
                                    float* sums, void* stream) {
  if (!xc || !yc || !sums || !ws || R <= 0 || n <= 0) return -1;
  const int g = grid_red(n);                                                 // <= RFX_STFT_REDUCE_SLOTS
  hipLaunchKernelGGL(stft_loss_reduce_kernel, dim3(g, R), dim3(256), 0, (hipStream_t)stream,
                     (const float2*)xc, (const float2*)yc, n, eps, ws);
  RFX_CHECK_LAUNCH();
  hipLaunchKernelGGL(rfx_slot_sum_kernel<float>, RFX_SLOT_SUM_GRID(3 * R), 0, (hipStream_t)stream, ws, R, g, 3, sums);
  RFX_CHECK_LAUNCH();
  return 0;
}
extern "C" int rfx_stft_loss_grad(const float* xc, const float* yc, int32_t R, int64_t n, float eps,
                                  const float* sums, float w_sc, float w_lm, const float* gup, float* gxc, void* stream) {
  if (!xc || !yc || !sums || !gxc || R <= 0 || n <= 0) return -1;
  hipLaunchKernelGGL(stft_loss_grad_kernel, dim3(grid_x(n), R), dim3(256), 0, (hipStream_t)stream,
                     (const float2*)xc, (const float2*)yc, (const float*)nullptr, n, eps, sums, w_sc, w_lm, gup, (float2*)gxc);
  RFX_CHECK_LAUNCH();
  return 0;
}
extern "C" int rfx_stft_loss_grad_m(const float* xc, const float* ymag, int32_t R, int64_t n, float eps,
                                    const float* sums, float w_sc, float w_lm, const float* gup, float* gxc, void* stream) {
  if (!xc || !ymag || !sums || !gxc || R <= 0 || n <= 0) return -1;
  hipLaunchKernelGGL(stft_loss_grad_kernel, dim3(grid_x(n), R), dim3(256), 0, (hipStream_t)stream,
                     (const float2*)xc, (const float2*)nullptr, ymag, n, eps, sums, w_sc, w_lm, gup, (float2*)gxc);
  RFX_CHECK_LAUNCH();
  return 0;
}
extern "C" int rfx_l1_grad(const float* a, const float* b, int64_t n, float w, const float* gup, float* g, void* stream) {
  if (!a || !b || !g || n <= 0) return -1;
  hipLaunchKernelGGL(l1_grad_kernel, dim3(grid_x(n) * 4), dim3(256), 0, (hipStream_t)stream, a, b, n, w, gup, g);
  RFX_CHECK_LAUNCH();
  return 0;
}
extern "C" int rfx_sisdr_sums(const float* x, const float* t, int32_t R, int64_t L, int64_t x_rs,
                              int64_t t_rs, double* ws, double* sums, void* stream) {
  if (!x || !t || !sums || !ws || R <= 0 || L <= 0) return -1;
  int g = grid_x(L);
  g = g > RFX_SISDR_SLOTS ? RFX_SISDR_SLOTS : g;
  hipLaunchKernelGGL(sisdr_sums_kernel, dim3(g, R), dim3(256), 0, (hipStream_t)stream, x, t, L, x_rs, t_rs, ws);
  RFX_CHECK_LAUNCH();
  hipLaunchKernelGGL(rfx_slot_sum_kernel<double>, RFX_SLOT_SUM_GRID(5 * R), 0, (hipStream_t)stream, ws, R, g, 5, sums);
  RFX_CHECK_LAUNCH();
  return 0;
}
extern "C" int rfx_mrstft_combine(const float* const* sums, const int64_t* n, int32_t nres, int32_t R, int32_t per_example_sc,
                                  float* out, void* stream) {
  if (!sums || !n || !out || nres <= 0 || nres > 8 || R <= 0) return -1;
  MrCombineArgs a;
  for (int k = 0; k < 8; ++k) { a.sums[k] = nullptr; a.n[k] = 1.0; }
  for (int k = 0; k < nres; ++k) {
    if (!sums[k] || n[k] <= 0) return -1;
    a.sums[k] = sums[k];
    a.n[k] = (double)n[k];
  }
  a.nres = nres; a.R = R; a.per_example = per_example_sc; a.out = out;
  hipLaunchKernelGGL(mrstft_combine_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
  RFX_CHECK_LAUNCH();
  return 0;
}
extern "C" int rfx_sisdr_finish(const double* sums, int32_t R, int64_t L, int32_t zero_mean, double eps, float* out, void* stream) {
  if (!sums || !out || R <= 0 || L <= 0) return -1;
  hipLaunchKernelGGL(sisdr_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, R, (double)L, zero_mean, eps, out);
  RFX_CHECK_LAUNCH();
  return 0;
}
