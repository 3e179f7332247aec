// GroupNorm fused with the activation that always follows it on the HDemucs path
// (torchaudio HDemucs via models.py:319):  GELU (enc/dec norm1/norm2 -> gelu),
// GLU (rewrite -> norm -> glu), and the DConv tail  res + scale[c] * glu(gn(x)).
// x: (N, C, S) contiguous, G groups; a group is one contiguous run of (C/G)*S floats.
// HBM-bound: forward = 2 reads of x (stats, apply; the second usually hits L2) + 1 write;
// backward re-materialises u = gn(x) from x + (mean, rstd) instead of storing it.
// (torch-ROCm's own group_norm backward returned a wrong weight gradient for
//  (1024, 2, 20) / G=1 on this stack, scripts/debug_dconv.py -- one more reason.)
#include "common.h"

enum { GN_NONE = 0, GN_GELU = 1, GN_GLU = 2, GN_GLU_SCALE_RES = 3, GN_RELU = 4 };
// bn != 0: BatchNorm statistics -- one (mean, rstd) per CHANNEL over (N, S) (classifier.py:271-272)

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
  int bn;
  int x16;              // bf16 STORAGE of x (forward input) and, in the backward pass, of dx as well (bf16 arithmetic mode)
};

// element pointers with widening loads / rounding stores: x[i], dx[i] = v keep their syntax for both storage types
template <typename XT> struct GnXPtr {
  const XT* p;
  __device__ __forceinline__ float operator[](int64_t i) const { return rfx_ld1(p + i); }
  __device__ __forceinline__ GnXPtr operator+(int64_t o) const { return GnXPtr{p + o}; }
};
template <typename XT> struct GnDxPtr {
  XT* p;
  struct Ref { XT* q; __device__ __forceinline__ void operator=(float v) const { rfx_st1(q, v); } };
  __device__ __forceinline__ Ref operator[](int64_t i) const { return Ref{p + i}; }
  __device__ __forceinline__ GnDxPtr operator+(int64_t o) const { return GnDxPtr{p + o}; }
};
template <typename XT> __device__ __forceinline__ GnXPtr<XT> gn_xp(const GnArgs& a) { return GnXPtr<XT>{reinterpret_cast<const XT*>(a.x)}; }
template <typename XT> __device__ __forceinline__ GnDxPtr<XT> gn_dxp(const GnArgs& a) { return GnDxPtr<XT>{reinterpret_cast<XT*>(a.y)}; }

__device__ __forceinline__ int gn_sidx(const GnArgs& a, int n, int ch) {
  return a.bn ? ch : n * a.G + ch / (a.C / a.G);
}

// Work decomposition of the two reductions: one WAVE per (n, channel, S-chunk of <= 4096
// samples).  A group can be 12 x 65536 floats with only N = 64 groups in flight (time
// branch DConv) or 768 x 128 floats with thousands of groups; a block-per-group mapping
// left the first case at 64 workgroups on 256 CUs (rocprof r1a: 30 % of the Demucs step).
constexpr int GN_CHUNK = 4096;

template <typename XT, bool SLOTTED = false>
__global__ __launch_bounds__(256) void gn_stats_kernel(const GnArgs a, double* __restrict__ sums, int nchunks) {
  const int lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nitems = (int64_t)a.N * a.C * nchunks;
  if (item >= nitems) return;
  const int sc = (int)(item % nchunks);
  const int64_t r = item / nchunks;           // n*C + ch
  const int ch = (int)(r % a.C), n = (int)(r / a.C);
  const XT* xr = reinterpret_cast<const XT*>(a.x) + r * a.S;
  const int64_t s0 = (int64_t)sc * GN_CHUNK, s1 = min(s0 + GN_CHUNK, (int64_t)a.S);
  float p = 0.f, q = 0.f;
  if ((a.S & 3) == 0) {      // rows are 16-byte aligned: one dwordx4 per lane
    for (int64_t s = s0 + 4 * lane; s < s1; s += 256) {
      const f32x4 v = rfx_ld4(xr + s);
      p += (v[0] + v[1]) + (v[2] + v[3]);
      q += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
  } else {
    for (int64_t s = s0 + lane; s < s1; s += 64) { const float v = rfx_ld1(xr + s); p += v; q += v * v; }
  }
  const double dp = rfx_wave_sum_d((double)p), dq = rfx_wave_sum_d((double)q);
  if (lane == 0) {
    const int g = gn_sidx(a, n, ch);
    if (SLOTTED) {
      // (group, chunk) -- BatchNorm: (channel, sample, chunk) -- is THIS wave's alone: a plain store into its own slot, summed in slot
      // order by gn_finalize_kernel.  No zero fill, no atomics: a fill followed by fp64 atomics lost contributions whenever a second
      // stream kept the machine busy, and the sum no longer depends on the order the waves finish in.
      const int64_t slot = a.bn ? ((int64_t)g * a.N + n) * nchunks + sc : (int64_t)g * nchunks + sc;
      sums[2 * slot] = dp;
      sums[2 * slot + 1] = dq;
    } else {
      atomicAdd(sums + 2 * g, dp);
      atomicAdd(sums + 2 * g + 1, dq);
    }
  }
}

// one wave per group: lane l adds the group's slots l, l + 64, ... in order, then the fixed butterfly (BatchNorm over a batch has
// samples x chunks slots per channel: hundreds; a single thread would walk them as dependent loads)
__global__ __launch_bounds__(256) void gn_finalize_kernel(const double* __restrict__ sums, float* __restrict__ mean,
                                                          float* __restrict__ rstd, int ngroups, double len, float eps, int slots) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= ngroups) return;
  double s1 = 0.0, s2 = 0.0;                     // slots > 1: partial sums spread over several addresses
  for (int k = lane; k < slots; k += 64) { s1 += sums[2 * ((int64_t)i * slots + k)]; s2 += sums[2 * ((int64_t)i * slots + k) + 1]; }
  s1 = rfx_wave_sum_d(s1); s2 = rfx_wave_sum_d(s2);
  if (lane != 0) return;
  const double m = s1 / len;
  double var = s2 / len - m * m;
  var = var > 0.0 ? var : 0.0;
  mean[i] = (float)m;
  rstd[i] = (float)(1.0 / sqrt(var + (double)eps));
}

__device__ __forceinline__ float gn_u(const GnArgs& a, int n, int ch, int64_t s) {
  const int g = gn_sidx(a, n, ch);
  const float xv = a.x[((int64_t)n * a.C + ch) * a.S + s];
  return (xv - a.mean[g]) * a.rstd[g] * a.gamma[ch] + a.beta[ch];
}

// V = 4: S % 4 == 0, every thread handles 4 consecutive samples of one (n, channel[-pair]) row
template <int V>
__global__ __launch_bounds__(256) void gn_apply_kernel(const GnArgs a) {
  const bool glu = a.mode == GN_GLU || a.mode == GN_GLU_SCALE_RES;
  const int Co = glu ? a.C / 2 : a.C;
  const int64_t SV = a.S / V;
  const int64_t total = (int64_t)a.N * Co * SV;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t s = (i % SV) * V;
    const int64_t r = i / SV;
    const int c = (int)(r % Co), n = (int)(r / Co);
    const int ga = gn_sidx(a, n, c);
    const float ma = a.mean[ga], ra = a.rstd[ga] * a.gamma[c], ba = a.beta[c];
    const float* xa = a.x + ((int64_t)n * a.C + c) * a.S + s;
    float va[V], vb[V], rs[V];
    if (V == 4) { const f32x4 t = *reinterpret_cast<const f32x4*>(xa); va[0] = t[0]; va[1] = t[1]; va[2] = t[2]; va[3] = t[3]; }
    else {
#pragma unroll
      for (int q = 0; q < V; ++q) va[q] = xa[q];
    }
    float mb = 0.f, rb = 0.f, bb = 0.f, sc = 0.f;
    if (glu) {
      const int gb = gn_sidx(a, n, c + Co);
      mb = a.mean[gb]; rb = a.rstd[gb] * a.gamma[c + Co]; bb = a.beta[c + Co];
      const float* xb = xa + (int64_t)Co * a.S;
      if (V == 4) { const f32x4 t = *reinterpret_cast<const f32x4*>(xb); vb[0] = t[0]; vb[1] = t[1]; vb[2] = t[2]; vb[3] = t[3]; }
      else {
#pragma unroll
        for (int q = 0; q < V; ++q) vb[q] = xb[q];
      }
      if (a.mode == GN_GLU_SCALE_RES) {
        sc = a.scale[c];
        if (V == 4) { const f32x4 t = *reinterpret_cast<const f32x4*>(a.res + r * a.S + s); rs[0] = t[0]; rs[1] = t[1]; rs[2] = t[2]; rs[3] = t[3]; }
        else {
#pragma unroll
          for (int q = 0; q < V; ++q) rs[q] = a.res[r * a.S + s + q];
        }
      }
    }
    float o[V];
#pragma unroll
    for (int q = 0; q < V; ++q) {
      float v = (va[q] - ma) * ra + ba;
      if (a.mode == GN_GELU) v = rfx_gelu(v);
      else if (a.mode == GN_RELU) v = v > 0.f ? v : 0.f;
      else if (glu) {
        v = v * rfx_sigmoid((vb[q] - mb) * rb + bb);
        if (a.mode == GN_GLU_SCALE_RES) v = rs[q] + sc * v;
      }
      o[q] = v;
    }
    if (V == 4) { f32x4 t; t[0] = o[0]; t[1] = o[1]; t[2] = o[2]; t[3] = o[3]; *reinterpret_cast<f32x4*>(a.y + r * a.S + s) = t; }
    else {
#pragma unroll
      for (int q = 0; q < V; ++q) a.y[r * a.S + s + q] = o[q];
    }
  }
}

// ---- backward ---------------------------------------------------------------------
// u = gn(x) is re-materialised from x + (mean, rstd).  For the GLU modes one work item
// owns the channel PAIR (co, co + C/2): x_a, x_b and gy are read once and both du's come
// out of the same sigmoid.  Per-(n, channel) sums go to a small partial buffer (plain
// stores; atomics only when S is split into chunks) and two tiny kernels reduce it over
// channels (group sums) and over samples (dgamma / dbeta / dscale): no hot atomics.
struct GnDu { float du_a, xh_a, du_b, xh_b, gf; };

template <typename XT = float>
__device__ __forceinline__ GnDu gn_du_pair(const GnArgs& a, int n, int co, int64_t s) {
  GnDu r;
  const int Co = a.C / 2;
  const int ga = gn_sidx(a, n, co), gb = gn_sidx(a, n, co + Co);
  const GnXPtr<XT> X = gn_xp<XT>(a);
  const float xa = X[((int64_t)n * a.C + co) * a.S + s], xb = X[((int64_t)n * a.C + co + Co) * a.S + s];
  r.xh_a = (xa - a.mean[ga]) * a.rstd[ga];
  r.xh_b = (xb - a.mean[gb]) * a.rstd[gb];
  const float ua = r.xh_a * a.gamma[co] + a.beta[co], ub = r.xh_b * a.gamma[co + Co] + a.beta[co + Co];
  float g0 = a.gy[((int64_t)n * Co + co) * a.S + s];
  const float sg = rfx_sigmoid(ub);
  r.gf = 0.f;
  if (a.mode == GN_GLU_SCALE_RES) { r.gf = g0 * ua * sg; g0 *= a.scale[co]; }
  r.du_a = g0 * sg;
  r.du_b = g0 * ua * sg * (1.f - sg);
  return r;
}

template <typename XT = float>
__device__ __forceinline__ float gn_du_single(const GnArgs& a, int n, int ch, int64_t s, float& xhat) {
  const int g = gn_sidx(a, n, ch);
  const int64_t i = ((int64_t)n * a.C + ch) * a.S + s;
  xhat = (gn_xp<XT>(a)[i] - a.mean[g]) * a.rstd[g];
  const float g0 = a.gy[i];
  if (a.mode == GN_GELU) return g0 * rfx_gelu_grad(xhat * a.gamma[ch] + a.beta[ch]);
  if (a.mode == GN_RELU) return (xhat * a.gamma[ch] + a.beta[ch]) > 0.f ? g0 : 0.f;
  return g0;
}

// part: (N, C, 2) = { sum du, sum du*xhat } per (n, channel);  psc: (N, C/2) = sum gy*f (mode 3)
// nchunks > 1: `part` / `psc` are SLOT arrays ((n, channel, chunk) is one wave's alone: plain stores), added in chunk order by
// gn_bwd_slotsum_kernel -- the fill + atomics form this replaced depended on the order the waves finished in
template <typename XT>
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const GnArgs a, float* __restrict__ part,
                                                             float* __restrict__ psc, int nchunks) {
  const int lane = threadIdx.x & 63;
  const bool pair = a.mode == GN_GLU || a.mode == GN_GLU_SCALE_RES;
  const int Cw = pair ? a.C / 2 : a.C;      // work channels
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nitems = (int64_t)a.N * Cw * nchunks;
  if (item >= nitems) return;
  const int sc = (int)(item % nchunks);
  const int64_t r = item / nchunks;
  const int cw = (int)(r % Cw), n = (int)(r / Cw);
  const int64_t s0 = (int64_t)sc * GN_CHUNK, s1 = min(s0 + GN_CHUNK, (int64_t)a.S);
  float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  if (pair && (a.S & 3) == 0) {
    // 4 consecutive samples per lane, the (n, channel) constants loaded once per wave: the element-at-a-time loop below issues
    // three scalar loads and ~10 constant fetches per element and ran at 0.33 of HBM on 16-bit x (r02b profile)
    const int Co = a.C / 2;
    const int ga = gn_sidx(a, n, cw), gb = gn_sidx(a, n, cw + Co);
    const float mea = a.mean[ga], ra = a.rstd[ga], meb = a.mean[gb], rb = a.rstd[gb];
    const float gma = a.gamma[cw], bta = a.beta[cw], gmb = a.gamma[cw + Co], btb = a.beta[cw + Co];
    const float scl = a.mode == GN_GLU_SCALE_RES ? a.scale[cw] : 1.f;
    const XT* xa = reinterpret_cast<const XT*>(a.x) + ((int64_t)n * a.C + cw) * a.S;
    const XT* xb = xa + (int64_t)Co * a.S;
    const float* gr = a.gy + ((int64_t)n * Co + cw) * a.S;
    for (int64_t s = s0 + 4 * lane; s < s1; s += 256) {
      const f32x4 va = rfx_ld4(xa + s), vb = rfx_ld4(xb + s), g4 = rfx_ld4(gr + s);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float xha = (va[q] - mea) * ra, xhb = (vb[q] - meb) * rb;
        const float ua = xha * gma + bta, ub = xhb * gmb + btb;
        const float sg = rfx_sigmoid(ub);
        float g0 = g4[q];
        if (a.mode == GN_GLU_SCALE_RES) { v[4] += g0 * ua * sg; g0 *= scl; }
        const float dua = g0 * sg, dub = g0 * ua * sg * (1.f - sg);
        v[0] += dua; v[1] += dua * xha; v[2] += dub; v[3] += dub * xhb;
      }
    }
  } else if (pair) {
    for (int64_t s = s0 + lane; s < s1; s += 64) {
      const GnDu d = gn_du_pair<XT>(a, n, cw, s);
      v[0] += d.du_a; v[1] += d.du_a * d.xh_a; v[2] += d.du_b; v[3] += d.du_b * d.xh_b; v[4] += d.gf;
    }
  } else if ((a.S & 3) == 0) {
    // single-channel modes (GroupNorm + GELU / ReLU / none), 4 consecutive samples per lane, the (n, channel) constants loaded once
    // per wave: the element-at-a-time loop below was VALU-bound (r03 SQ counters: ~44 VALU instructions per element, 0.72 VALU-busy)
    const int gi = gn_sidx(a, n, cw);
    const float me = a.mean[gi], rs = a.rstd[gi], gm = a.gamma[cw], bt = a.beta[cw];
    const int64_t base = ((int64_t)n * a.C + cw) * a.S;
    const XT* xr = reinterpret_cast<const XT*>(a.x) + base;
    const float* gr = a.gy + base;
    for (int64_t s = s0 + 4 * lane; s < s1; s += 256) {
      const f32x4 vx = rfx_ld4(xr + s), g4 = rfx_ld4(gr + s);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float xh = (vx[q] - me) * rs;
        const float u = xh * gm + bt;
        float du = g4[q];
        if (a.mode == GN_GELU) du *= rfx_gelu_grad(u);
        else if (a.mode == GN_RELU) du = u > 0.f ? du : 0.f;
        v[0] += du; v[1] += du * xh;
      }
    }
  } else {
    for (int64_t s = s0 + lane; s < s1; s += 64) {
      float xh;
      const float du = gn_du_single<XT>(a, n, cw, s, xh);
      v[0] += du; v[1] += du * xh;
    }
  }
#pragma unroll
  for (int q = 0; q < 5; ++q) v[q] = rfx_wave_sum(v[q]);
  if (lane != 0) return;
  float* pa = part + ((int64_t)n * a.C + cw) * 2;
  if (nchunks == 1) {
    pa[0] = v[0]; pa[1] = v[1];
    if (pair) {
      float* pb = part + ((int64_t)n * a.C + cw + Cw) * 2;
      pb[0] = v[2]; pb[1] = v[3];
      if (a.mode == GN_GLU_SCALE_RES) psc[(int64_t)n * Cw + cw] = v[4];
    }
  } else {
    float* sa = part + (((int64_t)n * a.C + cw) * nchunks + sc) * 2;
    sa[0] = v[0]; sa[1] = v[1];
    if (pair) {
      float* sb = part + (((int64_t)n * a.C + cw + Cw) * nchunks + sc) * 2;
      sb[0] = v[2]; sb[1] = v[3];
      if (a.mode == GN_GLU_SCALE_RES) psc[((int64_t)n * Cw + cw) * nchunks + sc] = v[4];
    }
  }
}

// part[n][ch][0..1] = sum over chunks of the slots (in chunk order); psc[n][cw] likewise
__global__ __launch_bounds__(256) void gn_bwd_slotsum_kernel(const float* __restrict__ pslot, const float* __restrict__ scslot, float* __restrict__ part,
                                                             float* __restrict__ psc, int64_t nrows, int64_t nsc, int nchunks) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < nrows) {
    float s0 = 0.f, s1 = 0.f;
    for (int k = 0; k < nchunks; ++k) { s0 += pslot[(i * nchunks + k) * 2]; s1 += pslot[(i * nchunks + k) * 2 + 1]; }
    part[2 * i] = s0; part[2 * i + 1] = s1;
  }
  if (i < nsc) {
    float s2 = 0.f;
    for (int k = 0; k < nchunks; ++k) s2 += scslot[i * nchunks + k];
    psc[i] = s2;
  }
}

// gsum[n*G+g] = sum_{ch in g} gamma[ch] * part[n][ch]      (one wave per group)
__global__ __launch_bounds__(256) void gn_bwd_groupsum_kernel(const GnArgs a, const float* __restrict__ part) {
  const int lane = threadIdx.x & 63;
  const int grp = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (grp >= a.N * a.G) return;
  const int n = grp / a.G, g = grp % a.G, Cg = a.C / a.G;
  float s1 = 0.f, s2 = 0.f;
  for (int cc = lane; cc < Cg; cc += 64) {
    const int ch = g * Cg + cc;
    const float gam = a.gamma[ch];
    s1 += gam * part[((int64_t)n * a.C + ch) * 2];
    s2 += gam * part[((int64_t)n * a.C + ch) * 2 + 1];
  }
  s1 = rfx_wave_sum(s1); s2 = rfx_wave_sum(s2);
  if (lane == 0) { a.gsum[2 * grp] = s1; a.gsum[2 * grp + 1] = s2; }
}

// dbeta[ch] = sum_n part[n][ch][0], dgamma[ch] = sum_n part[n][ch][1], dscale[co] = sum_n psc[n][co]
__global__ __launch_bounds__(256) void gn_bwd_chansum_kernel(const GnArgs a, const float* __restrict__ part,
                                                             const float* __restrict__ psc) {
  const int lane = threadIdx.x & 63;
  const int ch = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ch >= a.C) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  const bool sc = a.mode == GN_GLU_SCALE_RES && ch < a.C / 2;
  for (int n = lane; n < a.N; n += 64) {
    s0 += part[((int64_t)n * a.C + ch) * 2];
    s1 += part[((int64_t)n * a.C + ch) * 2 + 1];
    if (sc) s2 += psc[(int64_t)n * (a.C / 2) + ch];
  }
  s0 = rfx_wave_sum(s0); s1 = rfx_wave_sum(s1); s2 = rfx_wave_sum(s2);
  if (lane == 0) {
    a.dbeta[ch] = s0; a.dgamma[ch] = s1;
    if (sc) a.dscale[ch] = s2;
    if (a.bn) { a.gsum[2 * ch] = a.gamma[ch] * s0; a.gsum[2 * ch + 1] = a.gamma[ch] * s1; }
  }
}

// Many samples (freq-branch DConvs: N = 32768): the kernel above gives one wave per channel a serial walk over all N with
// channel-strided reads (90-100 us for 13 MB).  Split N over NS slices (partials [NS][C][3] in `scratch`), then add the slices up.
__global__ __launch_bounds__(256) void gn_bwd_chansum_split_kernel(const GnArgs a, const float* __restrict__ part,
                                                                   const float* __restrict__ psc, float* __restrict__ scratch,
                                                                   int NS) {
  const int lane = threadIdx.x & 63;
  const int ch = blockIdx.x * 4 + (threadIdx.x >> 6), sl = blockIdx.y;
  if (ch >= a.C) return;
  const int per = (a.N + NS - 1) / NS, n0 = sl * per, n1 = min(n0 + per, a.N);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  const bool sc = a.mode == GN_GLU_SCALE_RES && ch < a.C / 2;
  for (int n = n0 + lane; n < n1; n += 64) {
    s0 += part[((int64_t)n * a.C + ch) * 2];
    s1 += part[((int64_t)n * a.C + ch) * 2 + 1];
    if (sc) s2 += psc[(int64_t)n * (a.C / 2) + ch];
  }
  s0 = rfx_wave_sum(s0); s1 = rfx_wave_sum(s1); s2 = rfx_wave_sum(s2);
  if (lane == 0) {
    float* o = scratch + ((int64_t)sl * a.C + ch) * 3;
    o[0] = s0; o[1] = s1; o[2] = s2;
  }
}
__global__ void gn_bwd_chansum_final_kernel(const GnArgs a, const float* __restrict__ scratch, int NS) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= a.C) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (int sl = 0; sl < NS; ++sl) {
    const float* o = scratch + ((int64_t)sl * a.C + ch) * 3;
    s0 += o[0]; s1 += o[1]; s2 += o[2];
  }
  a.dbeta[ch] = s0; a.dgamma[ch] = s1;
  if (a.mode == GN_GLU_SCALE_RES && ch < a.C / 2) a.dscale[ch] = s2;
}

template <int V>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const GnArgs a) {
  const bool pair = a.mode == GN_GLU || a.mode == GN_GLU_SCALE_RES;
  const int Cw = pair ? a.C / 2 : a.C, Cg = a.C / a.G;
  const int64_t SV = a.S / V;
  const int64_t total = (int64_t)a.N * Cw * SV;
  const float inv = a.bn ? 1.f / ((float)a.N * (float)a.S) : 1.f / ((float)Cg * (float)a.S);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t s = (i % SV) * V;
    const int64_t r = i / SV;
    const int cw = (int)(r % Cw), n = (int)(r / Cw);
    if (pair) {
      const int ga = gn_sidx(a, n, cw), gb = gn_sidx(a, n, cw + Cw);
      const float ra = a.rstd[ga], rb = a.rstd[gb];
      const float m1a = a.gsum[2 * ga] * inv, m2a = a.gsum[2 * ga + 1] * inv;
      const float m1b = a.gsum[2 * gb] * inv, m2b = a.gsum[2 * gb + 1] * inv;
      const float gma = a.gamma[cw], gmb = a.gamma[cw + Cw];
#pragma unroll
      for (int q = 0; q < V; ++q) {
        const GnDu d = gn_du_pair(a, n, cw, s + q);
        a.y[((int64_t)n * a.C + cw) * a.S + s + q] = ra * (d.du_a * gma - m1a - d.xh_a * m2a);
        a.y[((int64_t)n * a.C + cw + Cw) * a.S + s + q] = rb * (d.du_b * gmb - m1b - d.xh_b * m2b);
      }
    } else {
      const int g = gn_sidx(a, n, cw);
      const float rg = a.rstd[g], m1 = a.gsum[2 * g] * inv, m2 = a.gsum[2 * g + 1] * inv, gm = a.gamma[cw];
#pragma unroll
      for (int q = 0; q < V; ++q) {
        float xh;
        const float du = gn_du_single(a, n, cw, s + q, xh);
        a.y[r * a.S + s + q] = rg * (du * gm - m1 - xh * m2);
      }
    }
  }
}

// ---- S % 4 == 0: wave-per-(row, 256-sample chunk) forms of the two apply kernels ------------------------------
// The grid-stride kernels above decode (n, channel, s) from a flat 64-bit index with two 64-bit divisions per
// 4 elements, which costs about as many VALU cycles as the memory traffic takes.  Here the decode is two 32-bit
// divisions of a wave-uniform item index, the per-channel constants are uniform (scalar) loads, and every lane moves
// 16-byte vectors.  Same arithmetic, same results.
struct GnRow { int n, c; int64_t s; bool ok; };

// Work decode WITHOUT integer divisions (two runtime u32 divisions per thread were a third of the instructions of these 4-element
// threads): grid = (chunk groups, channel groups, samples).  ipr >= 3 chunks of 256 samples per row: a workgroup's 4 waves take 4
// consecutive chunks of one (n, channel) row; ipr = 1 or 2: they take 4 / ipr consecutive channels of one sample.
__device__ __forceinline__ GnRow gn_row_item(const GnArgs& a, int Cw, uint32_t nitems, uint32_t ipr) {
  GnRow r;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t ck;
  if (ipr >= 3) { r.c = (int)blockIdx.y; ck = blockIdx.x * 4u + wave; }
  else if (ipr == 2) { r.c = (int)(blockIdx.y * 2u + (wave >> 1)); ck = wave & 1u; }
  else { r.c = (int)(blockIdx.y * 4u + wave); ck = 0; }
  r.n = (int)blockIdx.z;
  r.s = (int64_t)ck * 256 + (threadIdx.x & 63) * 4;
  r.ok = r.c < Cw && ck < ipr && r.s < a.S;
  return r;
}
static dim3 gn_row_grid(int N, int Cw, uint32_t ipr) {
  const unsigned nz = (unsigned)(N < 65535 ? N : 65535);
  if (ipr >= 3) return dim3((ipr + 3) / 4, (unsigned)Cw, nz);
  return dim3(1, (unsigned)((Cw * ipr + 3) / 4), nz);
}

__device__ __forceinline__ f32x4 gn_ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

template <typename XT>
__global__ __launch_bounds__(256) void gn_apply_rows_kernel(const GnArgs a, uint32_t nitems, uint32_t ipr) {
  const bool glu = a.mode == GN_GLU || a.mode == GN_GLU_SCALE_RES;
  const int Co = glu ? a.C / 2 : a.C;
  const GnRow it = gn_row_item(a, Co, nitems, ipr);
  if (!it.ok) return;
  const int c = it.c;
  for (int n = it.n; n < a.N; n += (int)gridDim.z) {       // gridDim.z = min(N, 65535)
  const int ga = gn_sidx(a, n, c);
  const float ma = a.mean[ga], ra = a.rstd[ga] * a.gamma[c], ba = a.beta[c];
  const XT* xa = reinterpret_cast<const XT*>(a.x) + ((int64_t)n * a.C + c) * a.S + it.s;
  const int64_t oi = ((int64_t)n * Co + c) * a.S + it.s;
  const f32x4 va = rfx_ld4(xa);
  f32x4 vb = {0.f, 0.f, 0.f, 0.f}, rs = {0.f, 0.f, 0.f, 0.f};
  float mb = 0.f, rb = 0.f, bb = 0.f, sc = 0.f;
  if (glu) {
    const int gb = gn_sidx(a, n, c + Co);
    mb = a.mean[gb]; rb = a.rstd[gb] * a.gamma[c + Co]; bb = a.beta[c + Co];
    vb = rfx_ld4(xa + (int64_t)Co * a.S);
    if (a.mode == GN_GLU_SCALE_RES) { sc = a.scale[c]; rs = gn_ld4(a.res + oi); }
  }
  f32x4 o;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float v = (va[q] - ma) * ra + ba;
    if (a.mode == GN_GELU) v = rfx_gelu(v);
    else if (a.mode == GN_RELU) v = v > 0.f ? v : 0.f;
    else if (glu) {
      v = v * rfx_sigmoid((vb[q] - mb) * rb + bb);
      if (a.mode == GN_GLU_SCALE_RES) v = rs[q] + sc * v;
    }
    o[q] = v;
  }
  *reinterpret_cast<f32x4*>(a.y + oi) = o;
  }
}

template <typename XT>
__global__ __launch_bounds__(256) void gn_bwd_apply_rows_kernel(const GnArgs a, uint32_t nitems, uint32_t ipr) {
  const bool pair = a.mode == GN_GLU || a.mode == GN_GLU_SCALE_RES;
  const int Cw = pair ? a.C / 2 : a.C, Cg = a.C / a.G;
  const GnRow it = gn_row_item(a, Cw, nitems, ipr);
  if (!it.ok) return;
  const int cw = it.c;
  const float inv = a.bn ? 1.f / ((float)a.N * (float)a.S) : 1.f / ((float)Cg * (float)a.S);
  for (int n = it.n; n < a.N; n += (int)gridDim.z) {       // gridDim.z = min(N, 65535)
  const int64_t xi = ((int64_t)n * a.C + cw) * a.S + it.s;
  const f32x4 g4 = gn_ld4(a.gy + ((int64_t)n * Cw + cw) * a.S + it.s);
  const XT* X = reinterpret_cast<const XT*>(a.x);
  XT* DX = reinterpret_cast<XT*>(a.y);
  const f32x4 xa = rfx_ld4(X + xi);
  if (pair) {
    const f32x4 xb = rfx_ld4(X + xi + (int64_t)Cw * a.S);
    const int ga = gn_sidx(a, n, cw), gb = gn_sidx(a, n, cw + Cw);
    const float mea = a.mean[ga], meb = a.mean[gb], ra = a.rstd[ga], rb = a.rstd[gb];
    const float m1a = a.gsum[2 * ga] * inv, m2a = a.gsum[2 * ga + 1] * inv;
    const float m1b = a.gsum[2 * gb] * inv, m2b = a.gsum[2 * gb + 1] * inv;
    const float gma = a.gamma[cw], gmb = a.gamma[cw + Cw], bta = a.beta[cw], btb = a.beta[cw + Cw];
    const float sc = a.mode == GN_GLU_SCALE_RES ? a.scale[cw] : 1.f;
    f32x4 da, db;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float xha = (xa[q] - mea) * ra, xhb = (xb[q] - meb) * rb;
      const float ua = xha * gma + bta, ub = xhb * gmb + btb;
      const float sg = rfx_sigmoid(ub);
      const float g0 = a.mode == GN_GLU_SCALE_RES ? g4[q] * sc : g4[q];
      const float dua = g0 * sg, dub = g0 * ua * sg * (1.f - sg);
      da[q] = ra * (dua * gma - m1a - xha * m2a);
      db[q] = rb * (dub * gmb - m1b - xhb * m2b);
    }
    rfx_st4(DX + xi, da);
    rfx_st4(DX + xi + (int64_t)Cw * a.S, db);
  } else {
    const int g = gn_sidx(a, n, cw);
    const float me = a.mean[g], rg = a.rstd[g], m1 = a.gsum[2 * g] * inv, m2 = a.gsum[2 * g + 1] * inv;
    const float gm = a.gamma[cw], bt = a.beta[cw];
    f32x4 d;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float xh = (xa[q] - me) * rg;
      float du = g4[q];
      if (a.mode == GN_GELU) du = du * rfx_gelu_grad(xh * gm + bt);
      else if (a.mode == GN_RELU) du = (xh * gm + bt) > 0.f ? du : 0.f;
      d[q] = rg * (du * gm - m1 - xh * m2);
    }
    rfx_st4(DX + xi, d);
  }
  }
}

// rows x chunks of 256 samples; false if the item count does not fit the 32-bit decode
static bool gn_row_items(int64_t rows, int32_t S, uint32_t* nitems, uint32_t* ipr) {
  const int64_t per = ((int64_t)S + 255) / 256, n = rows * per;
  if ((S & 3) || n > 0x7fffffffLL) return false;
  *nitems = (uint32_t)n; *ipr = (uint32_t)per;
  return true;
}

static int gn_grid(int64_t total) {
  const int64_t b = (total + 1023) / 1024;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

static int norm_fwd(int x16, int bn, int use_given_stats, int sums_given_in, const float* x, const float* gamma, const float* beta, int32_t N, int32_t C,
                                 int32_t S, int32_t G, float eps, int32_t mode, const float* res,
                                 const float* scale, double* sums /* N*G*2 workspace */, float* mean,
                                 float* rstd, float* y, void* stream) {
  if (!x || !gamma || !beta || !mean || !rstd || !y || N <= 0 || C <= 0 || S <= 0 || G <= 0 || C % G) return -1;
  const bool sums_slotted = sums_given_in == -1;          // -1: no statistics given, `sums` is the slotted workspace (one pair per chunk)
  const int sums_given = sums_slotted ? 0 : sums_given_in;
  const bool glu = mode == GN_GLU || mode == GN_GLU_SCALE_RES;
  if (glu && (C % 2)) return -1;
  if (mode == GN_GLU_SCALE_RES && (!res || !scale)) return -1;
  GnArgs a{};
  a.x = x; a.gamma = gamma; a.beta = beta; a.mean = mean; a.rstd = rstd; a.y = y; a.res = res; a.scale = scale;
  a.N = N; a.C = C; a.S = S; a.G = G; a.mode = mode; a.eps = eps; a.bn = bn; a.x16 = x16;
  if (x16 && ((S & 3) || bn)) return -1;                   // bf16 storage: 8-byte vectors of 4 values, GroupNorm only
  const int nstat = bn ? C : N * G;
  hipStream_t s = (hipStream_t)stream;
  if (!use_given_stats) {
    if (!sums) return -1;
    // GroupNorm: a group is ONE contiguous run of (C / G) * S values, so the statistics kernel sees the tensor as (N, G, L) with one
    // "channel" per group and cuts L into 4096-value chunks.  (One wave per (n, channel) row left the deep layers -- C = 3072,
    // S = 128 -- with 196608 waves of 128 values and 768 same-address fp64 atomics per group: 0.06 of HBM in the r02c profile.)
    GnArgs st = a;
    if (!bn) { st.C = G; st.S = (C / G) * S; }
    const int nchunks = (st.S + GN_CHUNK - 1) / GN_CHUNK;
    const int64_t nitems = (int64_t)N * st.C * nchunks;
    int slots = sums_given > 1 ? sums_given : 1;
    if (!sums_given) {
      if (sums_given == 0 && (sums_slotted || bn)) {
        // `sums` holds N * G * nchunks pairs (rfx_groupnorm_stat_chunks; BatchNorm: C * N * nchunks, rfx_batchnorm_stat_slots): every
        // (group, chunk) wave stores its own
        if (x16) hipLaunchKernelGGL((gn_stats_kernel<rfx_bf16s, true>), dim3((unsigned)((nitems + 3) / 4)), dim3(256), 0, s, st, sums, nchunks);
        else hipLaunchKernelGGL((gn_stats_kernel<float, true>), dim3((unsigned)((nitems + 3) / 4)), dim3(256), 0, s, st, sums, nchunks);
        slots = bn ? N * nchunks : nchunks;
      } else {
        if (hipMemsetAsync(sums, 0, sizeof(double) * 2 * nstat, s) != hipSuccess) return -3;
        if (x16) hipLaunchKernelGGL(gn_stats_kernel<rfx_bf16s>, dim3((unsigned)((nitems + 3) / 4)), dim3(256), 0, s, st, sums, nchunks);
        else hipLaunchKernelGGL(gn_stats_kernel<float>, dim3((unsigned)((nitems + 3) / 4)), dim3(256), 0, s, st, sums, nchunks);
      }
      RFX_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((nstat + 3) / 4), dim3(256), 0, s, sums, mean, rstd, nstat,
                       bn ? (double)N * (double)S : (double)(C / G) * (double)S, eps, slots);
    RFX_CHECK_LAUNCH();
  }
  const int64_t total = (int64_t)N * (glu ? C / 2 : C) * S;
  uint32_t nrow_items = 0, ipr = 0;
  if (C <= 65535 && gn_row_items((int64_t)N * (glu ? C / 2 : C), S, &nrow_items, &ipr)) {
    const dim3 rg = gn_row_grid(N, glu ? C / 2 : C, ipr);
    if (x16) hipLaunchKernelGGL(gn_apply_rows_kernel<rfx_bf16s>, rg, dim3(256), 0, s, a, nrow_items, ipr);
    else hipLaunchKernelGGL(gn_apply_rows_kernel<float>, rg, dim3(256), 0, s, a, nrow_items, ipr);
  } else if (x16) return -1;
  else if ((S & 3) == 0) hipLaunchKernelGGL(gn_apply_kernel<4>, dim3(gn_grid(total / 4)), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(gn_apply_kernel<1>, dim3(gn_grid(total)), dim3(256), 0, s, a);
  RFX_CHECK_LAUNCH();
  return 0;
}

extern "C" int rfx_groupnorm_fwd(const float* x, const float* gamma, const float* beta, int32_t N, int32_t C,
                                 int32_t S, int32_t G, float eps, int32_t mode, const float* res,
                                 const float* scale, double* sums, int32_t sums_given, float* mean, float* rstd,
                                 float* y, void* stream) {
  return norm_fwd(0, 0, 0, sums_given, x, gamma, beta, N, C, S, G, eps, mode, res, scale, sums, mean, rstd, y, stream);
}
// x stored as bf16 (a conv output of the bf16 arithmetic mode); y, res fp32.  S % 4 == 0.
extern "C" int rfx_groupnorm_fwd_x16(const void* x, const float* gamma, const float* beta, int32_t N, int32_t C,
                                     int32_t S, int32_t G, float eps, int32_t mode, const float* res,
                                     const float* scale, double* sums, int32_t sums_given, float* mean, float* rstd,
                                     float* y, void* stream) {
  return norm_fwd(1, 0, 0, sums_given, static_cast<const float*>(x), gamma, beta, N, C, S, G, eps, mode, res, scale, sums, mean,
                  rstd, y, stream);
}
// BatchNorm over (N, S) per channel.  use_given_stats: mean / rstd are inputs (eval mode:
// running_mean, 1/sqrt(running_var + eps)); else batch statistics are computed and written.
extern "C" int rfx_batchnorm_fwd(const float* x, const float* gamma, const float* beta, int32_t N, int32_t C,
                                 int32_t S, float eps, int32_t mode, int32_t use_given_stats, double* sums,
                                 float* mean, float* rstd, float* y, void* stream) {
  if (mode != GN_NONE && mode != GN_RELU) return -1;
  return norm_fwd(0, 1, use_given_stats, 0, x, gamma, beta, N, C, S, C, eps, mode, nullptr, nullptr, sums, mean, rstd, y,
                  stream);
}

// GroupNorm(1, C) backward for SMALL samples (the freq-branch DConvs normalise 32768 x (96 x 256) "samples"): one
// workgroup owns a sample and does both passes back to back -- channel sums + the sample's two group sums, a barrier,
// then dx -- so the operands are re-read from L2 / Infinity Cache instead of HBM, there is no per-element index decoding
// and the per-channel constants are loaded once per row.  The wave-per-(n, channel, chunk) kernels above spend most of
// their time on row overhead at S = 256 (2.5 TB/s measured) and stay the path for large samples.
template <typename XT>
__global__ __launch_bounds__(256) void gn_bwd_sample_kernel(const GnArgs a, float* __restrict__ part,
                                                            float* __restrict__ psc) {
  __shared__ float red[4][2];
  const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool pair = a.mode == GN_GLU || a.mode == GN_GLU_SCALE_RES;
  const int Cw = pair ? a.C / 2 : a.C;
  const float mean = a.mean[n], rstd = a.rstd[n];
  const GnXPtr<XT> xn = gn_xp<XT>(a) + (int64_t)n * a.C * a.S;
  const float* gyn = a.gy + (int64_t)n * Cw * a.S;
  const GnDxPtr<XT> dxn = gn_dxp<XT>(a) + (int64_t)n * a.C * a.S;
  float gs1 = 0.f, gs2 = 0.f;
  // du of one element; pair modes return both halves (same formulas as gn_du_pair / gn_du_single)
  auto du_pair = [&](float xa, float xb, float g0, float ga, float ba, float gb, float bb, float sc, float& xha,
                     float& xhb, float& dua, float& dub, float& gf) {
    xha = (xa - mean) * rstd; xhb = (xb - mean) * rstd;
    const float ua = xha * ga + ba, ub = xhb * gb + bb;
    const float sg = rfx_sigmoid(ub);
    gf = 0.f;
    if (a.mode == GN_GLU_SCALE_RES) { gf = g0 * ua * sg; g0 *= sc; }
    dua = g0 * sg;
    dub = g0 * ua * sg * (1.f - sg);
  };
  auto du_single = [&](float xv, float g0, float gm, float bt, float& xh) {
    xh = (xv - mean) * rstd;
    if (a.mode == GN_GELU) return g0 * rfx_gelu_grad(xh * gm + bt);
    if (a.mode == GN_RELU) return (xh * gm + bt) > 0.f ? g0 : 0.f;
    return g0;
  };
  for (int cw = wave; cw < Cw; cw += 4) {
    float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (pair) {
      const float ga = a.gamma[cw], ba = a.beta[cw], gb = a.gamma[cw + Cw], bb = a.beta[cw + Cw];
      const float sc = a.mode == GN_GLU_SCALE_RES ? a.scale[cw] : 1.f;
      const GnXPtr<XT> xa = xn + (int64_t)cw * a.S;
      const GnXPtr<XT> xb = xn + (int64_t)(cw + Cw) * a.S;
      const float* gr = gyn + (int64_t)cw * a.S;
      for (int s = lane; s < a.S; s += 64) {
        float xha, xhb, dua, dub, gf;
        du_pair(xa[s], xb[s], gr[s], ga, ba, gb, bb, sc, xha, xhb, dua, dub, gf);
        v[0] += dua; v[1] += dua * xha; v[2] += dub; v[3] += dub * xhb; v[4] += gf;
      }
#pragma unroll
      for (int q = 0; q < 5; ++q) v[q] = rfx_wave_sum(v[q]);
      gs1 += ga * v[0] + gb * v[2];
      gs2 += ga * v[1] + gb * v[3];
      if (lane == 0) {
        float* pa = part + ((int64_t)n * a.C + cw) * 2;
        float* pb = part + ((int64_t)n * a.C + cw + Cw) * 2;
        pa[0] = v[0]; pa[1] = v[1]; pb[0] = v[2]; pb[1] = v[3];
        if (a.mode == GN_GLU_SCALE_RES) psc[(int64_t)n * Cw + cw] = v[4];
      }
    } else {
      const float gm = a.gamma[cw], bt = a.beta[cw];
      const GnXPtr<XT> xr = xn + (int64_t)cw * a.S;
      const float* gr = gyn + (int64_t)cw * a.S;
      for (int s = lane; s < a.S; s += 64) {
        float xh;
        const float du = du_single(xr[s], gr[s], gm, bt, xh);
        v[0] += du; v[1] += du * xh;
      }
      v[0] = rfx_wave_sum(v[0]); v[1] = rfx_wave_sum(v[1]);
      gs1 += gm * v[0];
      gs2 += gm * v[1];
      if (lane == 0) {
        float* pa = part + ((int64_t)n * a.C + cw) * 2;
        pa[0] = v[0]; pa[1] = v[1];
      }
    }
  }
  if (lane == 0) { red[wave][0] = gs1; red[wave][1] = gs2; }
  __syncthreads();
  const float inv = 1.f / ((float)a.C * (float)a.S);
  const float m1 = (red[0][0] + red[1][0] + red[2][0] + red[3][0]) * inv;
  const float m2 = (red[0][1] + red[1][1] + red[2][1] + red[3][1]) * inv;
  for (int cw = wave; cw < Cw; cw += 4) {
    if (pair) {
      const float ga = a.gamma[cw], ba = a.beta[cw], gb = a.gamma[cw + Cw], bb = a.beta[cw + Cw];
      const float sc = a.mode == GN_GLU_SCALE_RES ? a.scale[cw] : 1.f;
      const GnXPtr<XT> xa = xn + (int64_t)cw * a.S;
      const GnXPtr<XT> xb = xn + (int64_t)(cw + Cw) * a.S;
      const float* gr = gyn + (int64_t)cw * a.S;
      const GnDxPtr<XT> da = dxn + (int64_t)cw * a.S;
      const GnDxPtr<XT> db = dxn + (int64_t)(cw + Cw) * a.S;
      for (int s = lane; s < a.S; s += 64) {
        float xha, xhb, dua, dub, gf;
        du_pair(xa[s], xb[s], gr[s], ga, ba, gb, bb, sc, xha, xhb, dua, dub, gf);
        da[s] = rstd * (dua * ga - m1 - xha * m2);
        db[s] = rstd * (dub * gb - m1 - xhb * m2);
      }
    } else {
      const float gm = a.gamma[cw], bt = a.beta[cw];
      const GnXPtr<XT> xr = xn + (int64_t)cw * a.S;
      const float* gr = gyn + (int64_t)cw * a.S;
      const GnDxPtr<XT> dr = dxn + (int64_t)cw * a.S;
      for (int s = lane; s < a.S; s += 64) {
        float xh;
        const float du = du_single(xr[s], gr[s], gm, bt, xh);
        dr[s] = rstd * (du * gm - m1 - xh * m2);
      }
    }
  }
}

// Register-resident form of gn_bwd_sample_kernel for the GLU modes: 16 waves per sample, a wave owns RW channel pairs and
// keeps their (x_a, x_b, gy) rows -- S <= 64 * SV values each -- in registers between the statistics pass and the dx
// pass, so every operand is read exactly once (the 256-thread kernel re-reads them from L2 / Infinity Cache).
// KEEPG = false (RW = 6: up to 96 channel pairs): gy is streamed in both passes (its second read comes from cache) and only
// the two x rows stay in registers -- (x_a, x_b, gy) for six rows would spill at the 128 VGPRs a 1024-thread group has.
template <int RW, int SV, bool KEEPG = true, typename XT = float>
__global__ __launch_bounds__(1024) void gn_bwd_sample_reg_kernel(const GnArgs a, float* __restrict__ part,
                                                                 float* __restrict__ psc) {
  __shared__ float red[16][2];
  const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Cw = a.C / 2;
  const float mean = a.mean[n], rstd = a.rstd[n];
  const GnXPtr<XT> xn = gn_xp<XT>(a) + (int64_t)n * a.C * a.S;
  const float* gyn = a.gy + (int64_t)n * Cw * a.S;
  const GnDxPtr<XT> dxn = gn_dxp<XT>(a) + (int64_t)n * a.C * a.S;
  // 16-bit x: a lane owns PAIRS of adjacent samples (s = 2 lane + (q & 1) + 128 (q >> 1)) so that x and dx move as dwords (a 2-byte
  // access per lane makes every load / store instruction carry half a cache line: 3.3 TB/s in the r02b profile); S is even there
  constexpr bool PAIRED = sizeof(XT) == 2;
  auto sof = [&](int q) { return PAIRED ? 2 * lane + (q & 1) + 128 * (q >> 1) : lane + 64 * q; };
  float xa[RW][SV], xb[RW][SV], gg[KEEPG ? RW : 1][SV];
  // the sigmoid and the normalised first half are kept between the statistics pass and the dx pass where registers allow (RW = 3):
  // the kernel is VALU-bound (two passes of ~50 instructions per pair), not HBM-bound
  constexpr bool CACHE = RW <= 3;
  float sgc[CACHE ? RW : 1][SV], uac[CACHE ? RW : 1][SV];
#pragma unroll
  for (int j = 0; j < RW; ++j) {
    const int cw = wave + 16 * j;
    const int cc = cw < Cw ? cw : Cw - 1;                       // clamped: loads unconditional, results masked below
#pragma unroll
    for (int q = 0; q < SV; ++q) {
      const int s = sof(q);
      const int ss = s < a.S ? s : a.S - 1;
      if (PAIRED) {
        if ((q & 1) == 0) {                                      // one dword = samples (ss, ss + 1); ss is even, S is even
          const int s2 = s < a.S ? s : a.S - 2;
          const uint32_t ua = *reinterpret_cast<const uint32_t*>(xn.p + (int64_t)cc * a.S + s2);
          const uint32_t ub = *reinterpret_cast<const uint32_t*>(xn.p + (int64_t)(cc + Cw) * a.S + s2);
          xa[j][q] = __uint_as_float(ua << 16); xa[j][q + 1 < SV ? q + 1 : q] = __uint_as_float(ua & 0xffff0000u);
          xb[j][q] = __uint_as_float(ub << 16); xb[j][q + 1 < SV ? q + 1 : q] = __uint_as_float(ub & 0xffff0000u);
        }
      } else {
        xa[j][q] = xn[(int64_t)cc * a.S + ss];
        xb[j][q] = xn[(int64_t)(cc + Cw) * a.S + ss];
      }
      if (KEEPG) gg[j][q] = gyn[(int64_t)cc * a.S + ss];
    }
  }
  auto gval = [&](int j, int q, int cc) {
    if (KEEPG) return gg[j][q];
    const int s = sof(q);
    return gyn[(int64_t)cc * a.S + (s < a.S ? s : a.S - 1)];
  };
  float gs1 = 0.f, gs2 = 0.f;
#pragma unroll
  for (int j = 0; j < RW; ++j) {
    const int cw = wave + 16 * j;
    const bool cok = cw < Cw;
    const int cc = cok ? cw : Cw - 1;
    const float ga = a.gamma[cc], ba = a.beta[cc], gb = a.gamma[cc + Cw], bb = a.beta[cc + Cw];
    const float sc = a.mode == GN_GLU_SCALE_RES ? a.scale[cc] : 1.f;
    float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < SV; ++q) {
      const float ok = (cok && sof(q) < a.S) ? 1.f : 0.f;
      const float xha = (xa[j][q] - mean) * rstd, xhb = (xb[j][q] - mean) * rstd;
      const float ua = xha * ga + ba, ub = xhb * gb + bb;
      const float sg = rfx_sigmoid(ub);
      if (CACHE) { sgc[j][q] = sg; uac[j][q] = ua; }
      float g0 = gval(j, q, cc) * ok;
      float gf = 0.f;
      if (a.mode == GN_GLU_SCALE_RES) { gf = g0 * ua * sg; g0 *= sc; }
      const float dua = g0 * sg, dub = g0 * ua * sg * (1.f - sg);
      v[0] += dua; v[1] += dua * xha; v[2] += dub; v[3] += dub * xhb; v[4] += gf;
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) v[q] = rfx_wave_sum(v[q]);
    gs1 += ga * v[0] + gb * v[2];
    gs2 += ga * v[1] + gb * v[3];
    if (lane == 0 && cok) {
      float* pa = part + ((int64_t)n * a.C + cw) * 2;
      float* pb = part + ((int64_t)n * a.C + cw + Cw) * 2;
      pa[0] = v[0]; pa[1] = v[1]; pb[0] = v[2]; pb[1] = v[3];
      if (a.mode == GN_GLU_SCALE_RES) psc[(int64_t)n * Cw + cw] = v[4];
    }
  }
  if (lane == 0) { red[wave][0] = gs1; red[wave][1] = gs2; }
  __syncthreads();
  float t1 = 0.f, t2 = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) { t1 += red[k][0]; t2 += red[k][1]; }
  const float inv = 1.f / ((float)a.C * (float)a.S);
  const float m1 = t1 * inv, m2 = t2 * inv;
#pragma unroll
  for (int j = 0; j < RW; ++j) {
    const int cw = wave + 16 * j;
    if (cw >= Cw) continue;
    const float ga = a.gamma[cw], ba = a.beta[cw], gb = a.gamma[cw + Cw], bb = a.beta[cw + Cw];
    const float sc = a.mode == GN_GLU_SCALE_RES ? a.scale[cw] : 1.f;
    float da[SV], db[SV];
#pragma unroll
    for (int q = 0; q < SV; ++q) {
      const float xha = (xa[j][q] - mean) * rstd, xhb = (xb[j][q] - mean) * rstd;
      const float ua = CACHE ? uac[j][q] : xha * ga + ba;
      const float sg = CACHE ? sgc[j][q] : rfx_sigmoid(xhb * gb + bb);
      const float gq = gval(j, q, cw);
      const float g0 = a.mode == GN_GLU_SCALE_RES ? gq * sc : gq;
      const float dua = g0 * sg, dub = g0 * ua * sg * (1.f - sg);
      da[q] = rstd * (dua * ga - m1 - xha * m2);
      db[q] = rstd * (dub * gb - m1 - xhb * m2);
    }
#pragma unroll
    for (int q = 0; q < SV; ++q) {
      const int s = sof(q);
      if (PAIRED) {
        if ((q & 1) == 0 && s < a.S) {                           // (s, s + 1) as one dword
          const int q1 = q + 1 < SV ? q + 1 : q;
          *reinterpret_cast<uint32_t*>(dxn.p + (int64_t)cw * a.S + s) = rfx_bf16_bits(da[q]) | (rfx_bf16_bits(da[q1]) << 16);
          *reinterpret_cast<uint32_t*>(dxn.p + (int64_t)(cw + Cw) * a.S + s) = rfx_bf16_bits(db[q]) | (rfx_bf16_bits(db[q1]) << 16);
        }
      } else if (s < a.S) {
        dxn[(int64_t)cw * a.S + s] = da[q];
        dxn[(int64_t)(cw + Cw) * a.S + s] = db[q];
      }
    }
  }
}

// Tiny samples in the single-channel modes (DConv bottleneck: 12 / 24 channels x 256 frames): one WAVE owns a sample
// and holds it in registers -- wave reductions only, no workgroup barrier, operands read once.
template <int CMAX, int SV, typename XT = float>
__global__ __launch_bounds__(256) void gn_bwd_sample_wave_kernel(const GnArgs a, float* __restrict__ part) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= a.N) return;
  const float mean = a.mean[n], rstd = a.rstd[n];
  const GnXPtr<XT> xn = gn_xp<XT>(a) + (int64_t)n * a.C * a.S;
  const float* gyn = a.gy + (int64_t)n * a.C * a.S;
  const GnDxPtr<XT> dxn = gn_dxp<XT>(a) + (int64_t)n * a.C * a.S;
  float xv[CMAX][SV], gv[CMAX][SV];
#pragma unroll
  for (int c = 0; c < CMAX; ++c) {
    const int cc = c < a.C ? c : a.C - 1;
#pragma unroll
    for (int q = 0; q < SV; ++q) {
      const int s = lane + 64 * q;
      const int ss = s < a.S ? s : a.S - 1;
      xv[c][q] = xn[(int64_t)cc * a.S + ss];
      gv[c][q] = gyn[(int64_t)cc * a.S + ss];
    }
  }
  float gs1 = 0.f, gs2 = 0.f;
#pragma unroll
  for (int c = 0; c < CMAX; ++c) {
    const bool cok = c < a.C;
    const int cc = cok ? c : a.C - 1;
    const float gm = a.gamma[cc], bt = a.beta[cc];
    float v0 = 0.f, v1 = 0.f;
#pragma unroll
    for (int q = 0; q < SV; ++q) {
      const float xh = (xv[c][q] - mean) * rstd;
      float du = (cok && lane + 64 * q < a.S) ? gv[c][q] : 0.f;
      if (a.mode == GN_GELU) du *= rfx_gelu_grad(xh * gm + bt);
      else if (a.mode == GN_RELU) du = (xh * gm + bt) > 0.f ? du : 0.f;
      gv[c][q] = du;                               // keep du: the dx pass needs no second activation derivative
      v0 += du; v1 += du * xh;
    }
    v0 = rfx_wave_sum(v0); v1 = rfx_wave_sum(v1);
    gs1 += gm * v0;
    gs2 += gm * v1;
    if (lane == 0 && cok) {
      float* pa = part + ((int64_t)n * a.C + c) * 2;
      pa[0] = v0; pa[1] = v1;
    }
  }
  const float inv = 1.f / ((float)a.C * (float)a.S);
  const float m1 = gs1 * inv, m2 = gs2 * inv;
#pragma unroll
  for (int c = 0; c < CMAX; ++c) {
    if (c >= a.C) continue;
    const float gm = a.gamma[c];
#pragma unroll
    for (int q = 0; q < SV; ++q) {
      const int s = lane + 64 * q;
      const float xh = (xv[c][q] - mean) * rstd;
      if (s < a.S) dxn[(int64_t)c * a.S + s] = rstd * (gv[c][q] * gm - m1 - xh * m2);
    }
  }
}

static int norm_bwd(int x16, int bn, const float* x, const float* gamma, const float* beta, const float* mean,
                                 const float* rstd, const float* gy, int32_t N, int32_t C, int32_t S,
                                 int32_t G, int32_t mode, const float* scale,
                                 float* work /* rfx_norm_bwd_work_floats */, float* dx, float* dgamma,
                                 float* dbeta, float* dscale, void* stream) {
  if (!x || !gamma || !beta || !mean || !rstd || !gy || !work || !dx || !dgamma || !dbeta) return -1;
  if (N <= 0 || C <= 0 || S <= 0 || G <= 0 || C % G) return -1;
  const bool glu = mode == GN_GLU || mode == GN_GLU_SCALE_RES;
  if (glu && (C % 2)) return -1;
  if (mode == GN_GLU_SCALE_RES && (!scale || !dscale)) return -1;
  GnArgs a{};
  float* part = work;
  float* psc = work + (int64_t)N * C * 2;
  a.x = x; a.gamma = gamma; a.beta = beta; a.mean = const_cast<float*>(mean); a.rstd = const_cast<float*>(rstd);
  a.gy = gy; a.scale = scale; a.gsum = psc + (int64_t)N * (C / 2); a.y = dx; a.dgamma = dgamma; a.dbeta = dbeta;
  a.dscale = dscale;
  a.N = N; a.C = C; a.S = S; a.G = G; a.mode = mode; a.bn = bn; a.x16 = x16;
  if (x16 && ((S & 3) || bn)) return -1;
  hipStream_t s = (hipStream_t)stream;
  const int nchunks = (S + GN_CHUNK - 1) / GN_CHUNK;
  const int Cw = glu ? C / 2 : C;
  const int64_t nitems = (int64_t)N * Cw * nchunks;
  // nchunks > 1: the per-chunk slots live behind the group sums (rfx_norm_bwd_work_floats sizes `work`)
  float* pslot = a.gsum + (int64_t)2 * (bn ? C : (N * G > C ? N * G : C));
  float* scslot = pslot + (int64_t)N * C * 2 * nchunks;
  if (!bn && G == 1 && N >= 512 && (int64_t)C * S <= 65536) {
    // many small samples: one workgroup per sample, both passes fused (gn_bwd_sample_kernel); the GLU modes of the
    // HDemucs freq-branch shapes keep the sample in registers (gn_bwd_sample_reg_kernel)
#define GN_LAUNCH_X(KERNEL_F, KERNEL_H, ...)                                  \
  do {                                                                        \
    if (x16) hipLaunchKernelGGL(KERNEL_H, __VA_ARGS__);                       \
    else hipLaunchKernelGGL(KERNEL_F, __VA_ARGS__);                           \
  } while (0)
    if (!glu && S <= 256 && C <= 12)
      GN_LAUNCH_X((gn_bwd_sample_wave_kernel<12, 4, float>), (gn_bwd_sample_wave_kernel<12, 4, rfx_bf16s>), dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, a, part);
    else if (!glu && S <= 256 && C <= 24)
      GN_LAUNCH_X((gn_bwd_sample_wave_kernel<24, 4, float>), (gn_bwd_sample_wave_kernel<24, 4, rfx_bf16s>), dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, a, part);
    else if (glu && S <= 256 && C / 2 <= 48)
      GN_LAUNCH_X((gn_bwd_sample_reg_kernel<3, 4, true, float>), (gn_bwd_sample_reg_kernel<3, 4, true, rfx_bf16s>), dim3((unsigned)N), dim3(1024), 0, s, a, part, psc);
    else if (glu && S <= 256 && C / 2 <= 96)
      GN_LAUNCH_X((gn_bwd_sample_reg_kernel<6, 4, false, float>), (gn_bwd_sample_reg_kernel<6, 4, false, rfx_bf16s>), dim3((unsigned)N), dim3(1024), 0, s, a, part, psc);
    else
      GN_LAUNCH_X(gn_bwd_sample_kernel<float>, gn_bwd_sample_kernel<rfx_bf16s>, dim3((unsigned)N), dim3(256), 0, s, a, part, psc);
    RFX_CHECK_LAUNCH();
    // the group-sum region of `work` (2 N floats) is unused on this path: scratch for the sliced channel sums
    int NS = N >= 2048 ? 64 : 1;
    while (NS > 1 && (int64_t)NS * C * 3 > (int64_t)2 * N) NS >>= 1;
    if (NS > 1) {
      hipLaunchKernelGGL(gn_bwd_chansum_split_kernel, dim3((C + 3) / 4, NS), dim3(256), 0, s, a, part, psc, a.gsum, NS);
      RFX_CHECK_LAUNCH();
      hipLaunchKernelGGL(gn_bwd_chansum_final_kernel, dim3((C + 255) / 256), dim3(256), 0, s, a, a.gsum, NS);
    } else {
      hipLaunchKernelGGL(gn_bwd_chansum_kernel, dim3((C + 3) / 4), dim3(256), 0, s, a, part, psc);
    }
    RFX_CHECK_LAUNCH();
    return 0;
  }
  if (nchunks > 1) {
    GN_LAUNCH_X(gn_bwd_partial_kernel<float>, gn_bwd_partial_kernel<rfx_bf16s>, dim3((unsigned)((nitems + 3) / 4)), dim3(256), 0, s, a, pslot, scslot, nchunks);
    RFX_CHECK_LAUNCH();
    const int64_t nrows = (int64_t)N * C, nsc = (int64_t)N * (C / 2);
    hipLaunchKernelGGL(gn_bwd_slotsum_kernel, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, s, pslot, scslot, part, psc, nrows,
                       mode == GN_GLU_SCALE_RES ? nsc : 0, nchunks);
  } else {
    GN_LAUNCH_X(gn_bwd_partial_kernel<float>, gn_bwd_partial_kernel<rfx_bf16s>, dim3((unsigned)((nitems + 3) / 4)), dim3(256), 0, s, a, part, psc, nchunks);
  }
  RFX_CHECK_LAUNCH();
  if (!bn) {
    hipLaunchKernelGGL(gn_bwd_groupsum_kernel, dim3((N * G + 3) / 4), dim3(256), 0, s, a, part);
    RFX_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(gn_bwd_chansum_kernel, dim3((C + 3) / 4), dim3(256), 0, s, a, part, psc);
  RFX_CHECK_LAUNCH();
  uint32_t nrow_items = 0, ipr = 0;
  if (C <= 65535 && gn_row_items((int64_t)N * Cw, S, &nrow_items, &ipr))
    GN_LAUNCH_X(gn_bwd_apply_rows_kernel<float>, gn_bwd_apply_rows_kernel<rfx_bf16s>, gn_row_grid(N, Cw, ipr), dim3(256), 0, s, a, nrow_items, ipr);
  else if (x16) return -1;
  else if ((S & 3) == 0) hipLaunchKernelGGL(gn_bwd_apply_kernel<4>, dim3(gn_grid((int64_t)N * Cw * S / 4)), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(gn_bwd_apply_kernel<1>, dim3(gn_grid((int64_t)N * Cw * S)), dim3(256), 0, s, a);
  RFX_CHECK_LAUNCH();
  return 0;
}

extern "C" int rfx_groupnorm_bwd(const float* x, const float* gamma, const float* beta, const float* mean,
                                 const float* rstd, const float* gy, int32_t N, int32_t C, int32_t S,
                                 int32_t G, int32_t mode, const float* scale, float* work, float* dx,
                                 float* dgamma, float* dbeta, float* dscale, void* stream) {
  return norm_bwd(0, 0, x, gamma, beta, mean, rstd, gy, N, C, S, G, mode, scale, work, dx, dgamma, dbeta, dscale, stream);
}
// x and dx stored as bf16; gy fp32.  S % 4 == 0.
extern "C" int rfx_groupnorm_bwd_x16(const void* x, const float* gamma, const float* beta, const float* mean,
                                     const float* rstd, const float* gy, int32_t N, int32_t C, int32_t S,
                                     int32_t G, int32_t mode, const float* scale, float* work, void* dx,
                                     float* dgamma, float* dbeta, float* dscale, void* stream) {
  return norm_bwd(1, 0, static_cast<const float*>(x), gamma, beta, mean, rstd, gy, N, C, S, G, mode, scale, work,
                  static_cast<float*>(dx), dgamma, dbeta, dscale, stream);
}
// train-mode BatchNorm backward (batch statistics); work: rfx_norm_bwd_work_floats(N, C, S, 0) floats
extern "C" int rfx_batchnorm_bwd(const float* x, const float* gamma, const float* beta, const float* mean,
                                 const float* rstd, const float* gy, int32_t N, int32_t C, int32_t S,
                                 int32_t mode, float* work, float* dx, float* dgamma, float* dbeta, void* stream) {
  if (mode != GN_NONE && mode != GN_RELU) return -1;
  return norm_bwd(0, 1, x, gamma, beta, mean, rstd, gy, N, C, S, C, mode, nullptr, work, dx, dgamma, dbeta, nullptr, stream);
}

// ---- 2x2 average pooling (classifier.py:275) ---------------------------------------------
__global__ void avgpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t NC, int H, int W,
                                    int kh, int kw) {
  const int OH = H / kh, OW = W / kw;
  const int64_t total = NC * OH * OW;
  const float inv = 1.f / (float)(kh * kw);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ow = (int)(i % OW);
    const int64_t r = i / OW;
    const int oh = (int)(r % OH);
    const int64_t nc = r / OH;
    const float* p = x + (nc * H + (int64_t)oh * kh) * W + (int64_t)ow * kw;
    float acc = 0.f;
    for (int a = 0; a < kh; ++a)
      for (int b = 0; b < kw; ++b) acc += p[a * W + b];
    y[i] = acc * inv;
  }
}
__global__ void avgpool2_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, int64_t NC, int H, int W,
                                    int kh, int kw) {
  const int OH = H / kh, OW = W / kw;
  const int64_t total = NC * H * W;
  const float inv = 1.f / (float)(kh * kw);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int w = (int)(i % W);
    const int64_t r = i / W;
    const int h = (int)(r % H);
    const int64_t nc = r / H;
    const int oh = h / kh, ow = w / kw;
    gx[i] = (oh < OH && ow < OW) ? gy[(nc * OH + oh) * OW + ow] * inv : 0.f;
  }
}
extern "C" int rfx_avgpool2d_fwd(const float* x, float* y, int64_t NC, int32_t H, int32_t W, int32_t kh,
                                 int32_t kw, void* stream) {
  if (!x || !y || NC <= 0 || H < kh || W < kw || kh <= 0 || kw <= 0) return -1;
  hipLaunchKernelGGL(avgpool2_fwd_kernel, dim3(gn_grid(NC * (H / kh) * (W / kw))), dim3(256), 0, (hipStream_t)stream,
                     x, y, NC, H, W, kh, kw);
  RFX_CHECK_LAUNCH();
  return 0;
}
extern "C" int rfx_avgpool2d_bwd(const float* gy, float* gx, int64_t NC, int32_t H, int32_t W, int32_t kh,
                                 int32_t kw, void* stream) {
  if (!gy || !gx || NC <= 0 || H < kh || W < kw || kh <= 0 || kw <= 0) return -1;
  hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3(gn_grid(NC * H * W)), dim3(256), 0, (hipStream_t)stream, gy, gx, NC,
                     H, W, kh, kw);
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
// L = Co * S multiple of 4: one wave per (sample n, 256-element chunk of the (Co, S) half), 16-byte vectors, the only
// division is a wave-uniform 32-bit one (the flat kernels above divide 64-bit indices three times per element)
__global__ __launch_bounds__(256) void glu_fwd_rows_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           uint32_t nitems, uint32_t ipr, int64_t L) {
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t w = blockIdx.x * 4u + wave;
  if (w >= nitems) return;
  const uint32_t n = w / ipr, ck = w - n * ipr;
  const int64_t o = (int64_t)ck * 256 + (threadIdx.x & 63) * 4;
  if (o >= L) return;
  const f32x4 a = gn_ld4(x + (int64_t)n * 2 * L + o), b = gn_ld4(x + (int64_t)n * 2 * L + L + o);
  f32x4 r;
#pragma unroll
  for (int q = 0; q < 4; ++q) r[q] = a[q] * rfx_sigmoid(b[q]);
  *reinterpret_cast<f32x4*>(y + (int64_t)n * L + o) = r;
}
template <typename XT>      // XT: storage type of x (the conv output) and of gx (its gradient): float or rfx_bf16s
__global__ __launch_bounds__(256) void glu_bwd_rows_kernel(const XT* __restrict__ x, const float* __restrict__ gy,
                                                           XT* __restrict__ gx, uint32_t nitems, uint32_t ipr,
                                                           int64_t L) {
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t w = blockIdx.x * 4u + wave;
  if (w >= nitems) return;
  const uint32_t n = w / ipr, ck = w - n * ipr;
  const int64_t o = (int64_t)ck * 256 + (threadIdx.x & 63) * 4;
  if (o >= L) return;
  const int64_t ia = (int64_t)n * 2 * L + o;
  const f32x4 a = rfx_ld4(x + ia), b = rfx_ld4(x + ia + L), g = gn_ld4(gy + (int64_t)n * L + o);
  f32x4 da, db;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float sg = rfx_sigmoid(b[q]);
    da[q] = g[q] * sg;
    db[q] = g[q] * a[q] * sg * (1.f - sg);
  }
  rfx_st4(gx + ia, da);
  rfx_st4(gx + ia + L, db);
}
static bool glu_row_items(int64_t N, int64_t L, uint32_t* nitems, uint32_t* ipr) {
  const int64_t per = (L + 255) / 256;
  if ((L & 3) || N * per > 0x7fffffffLL) return false;
  *nitems = (uint32_t)(N * per); *ipr = (uint32_t)per;
  return true;
}
extern "C" int rfx_glu_fwd(const float* x, float* y, int64_t N, int64_t C, int64_t S, void* stream) {
  if (!x || !y || N <= 0 || C <= 0 || (C & 1) || S <= 0) return -1;
  uint32_t nitems = 0, ipr = 0;
  if (glu_row_items(N, (C / 2) * S, &nitems, &ipr))
    hipLaunchKernelGGL(glu_fwd_rows_kernel, dim3((nitems + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, y, nitems, ipr,
                       (C / 2) * S);
  else
    hipLaunchKernelGGL(glu_fwd_kernel, dim3(gn_grid(N * (C / 2) * S)), dim3(256), 0, (hipStream_t)stream, x, y, N, C / 2, S);
  RFX_CHECK_LAUNCH();
  return 0;
}
extern "C" int rfx_glu_bwd(const float* x, const float* gy, float* gx, int64_t N, int64_t C, int64_t S, void* stream) {
  if (!x || !gy || !gx || N <= 0 || C <= 0 || (C & 1) || S <= 0) return -1;
  uint32_t nitems = 0, ipr = 0;
  if (glu_row_items(N, (C / 2) * S, &nitems, &ipr))
    hipLaunchKernelGGL(glu_bwd_rows_kernel<float>, dim3((nitems + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, gy, gx, nitems,
                       ipr, (C / 2) * S);
  else
    hipLaunchKernelGGL(glu_bwd_kernel, dim3(gn_grid(N * (C / 2) * S)), dim3(256), 0, (hipStream_t)stream, x, gy, gx, N, C / 2, S);
  RFX_CHECK_LAUNCH();
  return 0;
}
// x (conv output) and gx (its gradient) stored as bf16 (bf16 mode: both are only ever GEMM operands / GLU inputs); gy fp32.
// (C / 2) * S must be a multiple of 4.
extern "C" int rfx_glu_bwd_bf16(const void* x, const float* gy, void* gx, int64_t N, int64_t C, int64_t S, void* stream) {
  if (!x || !gy || !gx || N <= 0 || C <= 0 || (C & 1) || S <= 0) return -1;
  uint32_t nitems = 0, ipr = 0;
  if (!glu_row_items(N, (C / 2) * S, &nitems, &ipr)) return -1;
  hipLaunchKernelGGL(glu_bwd_rows_kernel<rfx_bf16s>, dim3((nitems + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                     static_cast<const rfx_bf16s*>(x), gy, static_cast<rfx_bf16s*>(gx), nitems, ipr, (C / 2) * S);
  RFX_CHECK_LAUNCH();
  return 0;
}

// Chunks gn_stats_kernel cuts a group of a GroupNorm(G) over (C, S) into: a caller that passes sums_given = -1 to rfx_groupnorm_fwd
// / _x16 hands over a workspace of N * G * chunks pairs of doubles, filled by plain stores (no zero fill, no atomics).
// floats of `work` rfx_groupnorm_bwd[_x16] (G groups) / rfx_batchnorm_bwd (G = 0) need: per-(sample, channel) partial pairs, the
// LayerScale partials, the group (BatchNorm: channel) sums and, for rows longer than one 4096-sample chunk, the per-chunk slots
extern "C" int64_t rfx_norm_bwd_work_floats(int32_t N, int32_t C, int32_t S, int32_t G) {
  if (N <= 0 || C <= 0 || S <= 0 || G < 0) return -1;
  const int64_t nchunks = (S + GN_CHUNK - 1) / GN_CHUNK;
  const int64_t g = G == 0 ? C : ((int64_t)N * G > C ? (int64_t)N * G : C);
  int64_t n = (int64_t)N * C * 2 + (int64_t)N * (C / 2) + 2 * g;
  if (nchunks > 1) n += (int64_t)N * C * 2 * nchunks + (int64_t)N * (C / 2) * nchunks;
  return n;
}
// fp64 PAIRS of `sums` rfx_batchnorm_fwd needs per channel when it computes the batch statistics: one slot per (sample, chunk)
extern "C" int rfx_batchnorm_stat_slots(int32_t N, int32_t S) {
  if (N <= 0 || S <= 0) return -1;
  return N * ((S + GN_CHUNK - 1) / GN_CHUNK);
}
extern "C" int rfx_groupnorm_stat_chunks(int32_t C, int32_t S, int32_t G) {
  if (C <= 0 || S <= 0 || G <= 0 || C % G) return -1;
  const int64_t L = (int64_t)(C / G) * S;
  return (int)((L + GN_CHUNK - 1) / GN_CHUNK);
}
