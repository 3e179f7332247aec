// GroupNorm fused with the activation that always follows it on the HDemucs path
// (torchaudio HDemucs via models.py:319):  GELU (enc/dec norm1/norm2 -> gelu),
// GLU (rewrite -> norm -> glu), and the DConv tail  res + scale[c] * glu(gn(x)).
// x: (N, C, S) contiguous, G groups; a group is one contiguous run of (C/G)*S floats.
// HBM-bound: forward = 2 reads of x (stats, apply; the second usually hits L2) + 1 write;
// backward re-materialises u = gn(x) from x + (mean, rstd) instead of storing it.
// (torch-ROCm's own group_norm backward returned a wrong weight gradient for
//  (1024, 2, 20) / G=1 on this stack, scripts/debug_dconv.py -- one more reason.)
#include "common.h"

enum { GN_NONE = 0, GN_GELU = 1, GN_GLU = 2, GN_GLU_SCALE_RES = 3 };

struct GnArgs {
  const float* x;       // (N, C, S)
  const float* gamma;   // (C)
  const float* beta;    // (C)
  float* mean;          // (N*G)
  float* rstd;          // (N*G)
  float* y;             // fwd output / bwd: dx
  const float* res;     // mode 3: residual (N, C/2, S)
  const float* scale;   // mode 3: LayerScale (C/2)
  const float* gy;      // bwd: grad of the output
  float* dgamma;        // bwd (C), atomics
  float* dbeta;         // bwd (C)
  float* dscale;        // bwd mode 3 (C/2)
  float* gsum;          // bwd (N*G, 2): sum dxhat, sum dxhat*xhat
  int N, C, S, G, mode;
  float eps;
};

__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = rfx_wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void gn_stats_kernel(const GnArgs a) {
  __shared__ float sh[4];
  const int64_t len = (int64_t)(a.C / a.G) * a.S;
  const float* xg = a.x + (int64_t)blockIdx.x * len;
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < len; i += 256) s += xg[i];
  const float mean = block_sum(s, sh) / (float)len;
  float v = 0.f;
  for (int64_t i = threadIdx.x; i < len; i += 256) { const float d = xg[i] - mean; v += d * d; }
  const float var = block_sum(v, sh) / (float)len;
  if (threadIdx.x == 0) { a.mean[blockIdx.x] = mean; a.rstd[blockIdx.x] = rsqrtf(var + a.eps); }
}

__device__ __forceinline__ float gn_u(const GnArgs& a, int n, int ch, int64_t s) {
  const int g = ch / (a.C / a.G);
  const float xv = a.x[((int64_t)n * a.C + ch) * a.S + s];
  return (xv - a.mean[n * a.G + g]) * a.rstd[n * a.G + g] * a.gamma[ch] + a.beta[ch];
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const GnArgs a) {
  const int Co = a.mode >= GN_GLU ? a.C / 2 : a.C;
  const int64_t total = (int64_t)a.N * Co * a.S;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t s = i % a.S;
    const int64_t r = i / a.S;
    const int c = (int)(r % Co), n = (int)(r / Co);
    float v = gn_u(a, n, c, s);
    if (a.mode == GN_GELU) v = rfx_gelu(v);
    else if (a.mode >= GN_GLU) {
      v = v * rfx_sigmoid(gn_u(a, n, c + Co, s));
      if (a.mode == GN_GLU_SCALE_RES) v = a.res[i] + a.scale[c] * v;
    }
    a.y[i] = v;
  }
}

// du for input element (n, ch, s) given the output gradient; also returns xhat and (mode 3) g*f
__device__ __forceinline__ float gn_du(const GnArgs& a, int n, int ch, int64_t s, float& xhat, float& gf) {
  const int g = ch / (a.C / a.G);
  const float xv = a.x[((int64_t)n * a.C + ch) * a.S + s];
  xhat = (xv - a.mean[n * a.G + g]) * a.rstd[n * a.G + g];
  const float u = xhat * a.gamma[ch] + a.beta[ch];
  gf = 0.f;
  if (a.mode <= GN_GELU) {
    const float g0 = a.gy[((int64_t)n * a.C + ch) * a.S + s];
    return a.mode == GN_GELU ? g0 * rfx_gelu_grad(u) : g0;
  }
  const int Co = a.C / 2;
  const bool is_a = ch < Co;
  const int co = is_a ? ch : ch - Co;
  float g0 = a.gy[((int64_t)n * Co + co) * a.S + s];
  const float other = gn_u(a, n, is_a ? ch + Co : ch - Co, s);
  const float ua = is_a ? u : other, ub = is_a ? other : u;
  const float sg = rfx_sigmoid(ub);
  if (a.mode == GN_GLU_SCALE_RES) {
    if (is_a) gf = g0 * ua * sg;      // d/dscale, counted once (on the 'a' half)
    g0 *= a.scale[co];
  }
  return is_a ? g0 * sg : g0 * ua * sg * (1.f - sg);
}

// one block per (n, g); each wave walks whole channels so per-channel sums need no barrier
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const GnArgs a) {
  __shared__ float sh[2][4];
  const int n = blockIdx.x / a.G, g = blockIdx.x % a.G;
  const int Cg = a.C / a.G;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float s1 = 0.f, s2 = 0.f;
  for (int cc = wave; cc < Cg; cc += 4) {
    const int ch = g * Cg + cc;
    float dg = 0.f, db = 0.f, ds = 0.f;
    for (int64_t s = lane; s < a.S; s += 64) {
      float xhat, gf;
      const float du = gn_du(a, n, ch, s, xhat, gf);
      dg += du * xhat; db += du; ds += gf;
    }
    dg = rfx_wave_sum(dg); db = rfx_wave_sum(db);
    if (a.mode == GN_GLU_SCALE_RES) ds = rfx_wave_sum(ds);
    if (lane == 0) {
      atomicAdd(a.dgamma + ch, dg);
      atomicAdd(a.dbeta + ch, db);
      if (a.mode == GN_GLU_SCALE_RES && ch < a.C / 2) atomicAdd(a.dscale + ch, ds);
    }
    const float gam = a.gamma[ch];
    s1 += db * gam;      // sum dxhat       (dxhat = du * gamma)
    s2 += dg * gam;      // sum dxhat*xhat
  }
  if (lane == 0) { sh[0][wave] = s1; sh[1][wave] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    a.gsum[2 * blockIdx.x] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
    a.gsum[2 * blockIdx.x + 1] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
  }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const GnArgs a) {
  const int64_t total = (int64_t)a.N * a.C * a.S;
  const int Cg = a.C / a.G;
  const float inv = 1.f / ((float)Cg * (float)a.S);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t s = i % a.S;
    const int64_t r = i / a.S;
    const int ch = (int)(r % a.C), n = (int)(r / a.C);
    const int g = ch / Cg;
    float xhat, gf;
    const float du = gn_du(a, n, ch, s, xhat, gf);
    const float m1 = a.gsum[2 * (n * a.G + g)] * inv, m2 = a.gsum[2 * (n * a.G + g) + 1] * inv;
    a.y[i] = a.rstd[n * a.G + g] * (du * a.gamma[ch] - m1 - xhat * m2);
  }
}

static int gn_grid(int64_t total) {
  const int64_t b = (total + 1023) / 1024;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

extern "C" int rfx_groupnorm_fwd(const float* x, const float* gamma, const float* beta, int32_t N, int32_t C,
                                 int32_t S, int32_t G, float eps, int32_t mode, const float* res,
                                 const float* scale, float* mean, float* rstd, float* y, void* stream) {
  if (!x || !gamma || !beta || !mean || !rstd || !y || N <= 0 || C <= 0 || S <= 0 || G <= 0 || C % G) return -1;
  if (mode >= GN_GLU && (C % 2)) return -1;
  if (mode == GN_GLU_SCALE_RES && (!res || !scale)) return -1;
  GnArgs a{};
  a.x = x; a.gamma = gamma; a.beta = beta; a.mean = mean; a.rstd = rstd; a.y = y; a.res = res; a.scale = scale;
  a.N = N; a.C = C; a.S = S; a.G = G; a.mode = mode; a.eps = eps;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gn_stats_kernel, dim3(N * G), dim3(256), 0, s, a);
  RFX_CHECK_LAUNCH();
  const int64_t total = (int64_t)N * (mode >= GN_GLU ? C / 2 : C) * S;
  hipLaunchKernelGGL(gn_apply_kernel, dim3(gn_grid(total)), dim3(256), 0, s, a);
  RFX_CHECK_LAUNCH();
  return 0;
}

extern "C" int rfx_groupnorm_bwd(const float* x, const float* gamma, const float* beta, const float* mean,
                                 const float* rstd, const float* gy, int32_t N, int32_t C, int32_t S,
                                 int32_t G, int32_t mode, const float* scale, float* gsum /* N*G*2 */,
                                 float* dx, float* dgamma, float* dbeta, float* dscale, void* stream) {
  if (!x || !gamma || !beta || !mean || !rstd || !gy || !gsum || !dx || !dgamma || !dbeta) return -1;
  if (N <= 0 || C <= 0 || S <= 0 || G <= 0 || C % G) return -1;
  if (mode == GN_GLU_SCALE_RES && (!scale || !dscale)) return -1;
  GnArgs a{};
  a.x = x; a.gamma = gamma; a.beta = beta; a.mean = const_cast<float*>(mean); a.rstd = const_cast<float*>(rstd);
  a.gy = gy; a.scale = scale; a.gsum = gsum; a.y = dx; a.dgamma = dgamma; a.dbeta = dbeta; a.dscale = dscale;
  a.N = N; a.C = C; a.S = S; a.G = G; a.mode = mode;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3(N * G), dim3(256), 0, s, a);
  RFX_CHECK_LAUNCH();
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(gn_grid((int64_t)N * C * S)), dim3(256), 0, s, a);
  RFX_CHECK_LAUNCH();
  return 0;
}

// ---- plain GLU (layers without a norm) and a*x + b*y ------------------------------------
__global__ void glu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t N, int64_t Co, int64_t S) {
  const int64_t total = N * Co * S;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t s = i % S, r = i / S, c = r % Co, n = r / Co;
    const float a = x[(n * 2 * Co + c) * S + s], b = x[(n * 2 * Co + c + Co) * S + s];
    y[i] = a * rfx_sigmoid(b);
  }
}
__global__ void glu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ gx,
                               int64_t N, int64_t Co, int64_t S) {
  const int64_t total = N * Co * S;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t s = i % S, r = i / S, c = r % Co, n = r / Co;
    const int64_t ia = (n * 2 * Co + c) * S + s, ib = ia + Co * S;
    const float a = x[ia], sg = rfx_sigmoid(x[ib]), g = gy[i];
    gx[ia] = g * sg;
    gx[ib] = g * a * sg * (1.f - sg);
  }
}
extern "C" int rfx_glu_fwd(const float* x, float* y, int64_t N, int64_t C, int64_t S, void* stream) {
  if (!x || !y || N <= 0 || C <= 0 || (C & 1) || S <= 0) return -1;
  hipLaunchKernelGGL(glu_fwd_kernel, dim3(gn_grid(N * (C / 2) * S)), dim3(256), 0, (hipStream_t)stream, x, y, N, C / 2, S);
  RFX_CHECK_LAUNCH();
  return 0;
}
extern "C" int rfx_glu_bwd(const float* x, const float* gy, float* gx, int64_t N, int64_t C, int64_t S, void* stream) {
  if (!x || !gy || !gx || N <= 0 || C <= 0 || (C & 1) || S <= 0) return -1;
  hipLaunchKernelGGL(glu_bwd_kernel, dim3(gn_grid(N * (C / 2) * S)), dim3(256), 0, (hipStream_t)stream, x, gy, gx, N, C / 2, S);
  RFX_CHECK_LAUNCH();
  return 0;
}
