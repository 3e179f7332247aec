// LocalState attention of the Hybrid Demucs DConv blocks (torchaudio HDemucs `_LocalState`, reached from
// remfx/models.py:319): 4 heads over T <= 256 frames with a learned decay penalty and a masked diagonal,
//   dots[t, s] = <k[:, t], q[:, s]> / sqrt(ch) - sum_f (f + 1) |t - s| / sqrt(nd) * sigmoid(qd[f, s]) / 2,  dots[s, s] = -100
//   w = softmax over t;   out[c, s] = sum_t w[t, s] * content[c, t]
// One workgroup per (batch row, head); k and content (ch x T fp32 each, <= 48 KB) stay in LDS while the workgroup walks
// over blocks of 32 query columns s; the score block (T x 32) lives in LDS too.  Exact fp32 on the vector ALU: the
// whole op is 6 GFLOP per Demucs step (0.03 % of it), so this is about removing the rocBLAS / ATen softmax launches and
// the (B, h, T, T) round trips, not about the matrix pipe.  Forward + backward.
#include "common.h"

// SB = query columns per block (32 forward, 16 backward: two T x SB blocks + k + content must fit 160 KB of LDS);
// score-block rows are padded to SB + 1 floats

struct LsArgs {
  const float *q, *k, *cont, *qd;   // (B, heads*ch, T) x3, (B, heads*nd, T)
  float* w;                         // (B, heads, T, T) attention weights [t][s] (saved for the backward), may be null
  float* out;                       // (B, heads*ch, T)
  const float* gout;                // backward: d out
  float *dq, *dk, *dcont, *dqd;
  int heads, ch, T, nd;
};

__device__ __forceinline__ float ls_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// scores of one column block into sc[t][s] (raw, before softmax)
template <int LS_SB>
__device__ __forceinline__ void ls_scores(const LsArgs& a, const float* kk, const float* qs, const float* dql, float* sc,
                                          int s0, int tid) {
  constexpr int LS_LD = LS_SB + 1;
  const int T = a.T, ch = a.ch;
  const float inv = 1.0f / sqrtf((float)ch), invd = 1.0f / sqrtf((float)a.nd);
  for (int it = tid; it < T * (LS_SB / 4); it += 256) {
    const int t = it % T, s4 = (it / T) * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < ch; ++c) {
      const float kv = kk[c * T + t];
      const f32x4 qv = *reinterpret_cast<const f32x4*>(qs + c * LS_SB + s4);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = fmaf(kv, qv[j], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int s = s0 + s4 + j;
      float v = acc[j] * inv;
      const float dist = fabsf((float)(t - s));
      for (int f = 0; f < a.nd; ++f) v -= (float)(f + 1) * dist * invd * dql[f * LS_SB + s4 + j];
      sc[t * LS_LD + s4 + j] = (t == s) ? -100.0f : v;
    }
  }
}

// softmax over t of every column of sc (in place); red: 2 x NP x LS_SB floats of scratch
template <int LS_SB>
__device__ __forceinline__ void ls_softmax(float* sc, float* red, int T, int tid) {
  constexpr int LS_LD = LS_SB + 1, NP = 256 / LS_SB;   // NP row parts per column
  const int s = tid % LS_SB, p = tid / LS_SB;
  const int rows = (T + NP - 1) / NP, t0 = p * rows, t1 = min(t0 + rows, T);
  float m = -3.0e38f;
  for (int t = t0; t < t1; ++t) m = fmaxf(m, sc[t * LS_LD + s]);
  red[p * LS_SB + s] = m;
  __syncthreads();
  m = red[s];
#pragma unroll
  for (int i = 1; i < NP; ++i) m = fmaxf(m, red[i * LS_SB + s]);
  float sum = 0.f;
  for (int t = t0; t < t1; ++t) {
    const float e = expf(sc[t * LS_LD + s] - m);
    sc[t * LS_LD + s] = e;
    sum += e;
  }
  red[(NP + p) * LS_SB + s] = sum;
  __syncthreads();
  sum = 0.f;
#pragma unroll
  for (int i = 0; i < NP; ++i) sum += red[(NP + i) * LS_SB + s];
  const float r = 1.0f / sum;
  for (int t = t0; t < t1; ++t) sc[t * LS_LD + s] *= r;
  __syncthreads();
}

template <int LS_SB>
__global__ __launch_bounds__(256) void localstate_fwd_kernel(const LsArgs a) {
  constexpr int LS_LD = LS_SB + 1;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int T = a.T, ch = a.ch, tid = threadIdx.x;
  float* kk = lds;                       // [ch][T]
  float* cc = kk + ch * T;               // [ch][T]
  float* sc = cc + ch * T;               // [T][LS_LD]
  float* qs = sc + T * LS_LD;            // [ch][LS_SB]
  float* dql = qs + ch * LS_SB;          // [nd][LS_SB]  sigmoid(qd) / 2
  float* red = dql + a.nd * LS_SB;       // [2 * 256 / LS_SB][LS_SB] = 512 floats
  const int b = blockIdx.x / a.heads, hd = blockIdx.x % a.heads;
  const int64_t base = ((int64_t)b * a.heads + hd) * ch * T;
  const int64_t dbase = ((int64_t)b * a.heads + hd) * a.nd * T;
  for (int i = tid; i < ch * T; i += 256) { kk[i] = a.k[base + i]; cc[i] = a.cont[base + i]; }
  for (int s0 = 0; s0 < T; s0 += LS_SB) {
    __syncthreads();
    for (int i = tid; i < ch * LS_SB; i += 256) {
      const int c = i / LS_SB, s = i % LS_SB;
      qs[i] = s0 + s < T ? a.q[base + (int64_t)c * T + s0 + s] : 0.f;
    }
    for (int i = tid; i < a.nd * LS_SB; i += 256) {
      const int f = i / LS_SB, s = i % LS_SB;
      dql[i] = s0 + s < T ? 0.5f * ls_sigmoid(a.qd[dbase + (int64_t)f * T + s0 + s]) : 0.f;
    }
    __syncthreads();
    ls_scores<LS_SB>(a, kk, qs, dql, sc, s0, tid);
    __syncthreads();
    ls_softmax<LS_SB>(sc, red, T, tid);
    if (a.w) {
      float* wb = a.w + ((int64_t)b * a.heads + hd) * T * T;
      for (int i = tid; i < T * LS_SB; i += 256) {
        const int t = i / LS_SB, s = i % LS_SB;
        if (s0 + s < T) wb[(int64_t)t * T + s0 + s] = sc[t * LS_LD + s];
      }
    }
    // out[c][s] = sum_t w[t][s] * cont[c][t]
    for (int i = tid; i < ch * LS_SB; i += 256) {
      const int c = i / LS_SB, s = i % LS_SB;
      float acc = 0.f;
      for (int t = 0; t < T; ++t) acc = fmaf(sc[t * LS_LD + s], cc[c * T + t], acc);
      if (s0 + s < T) a.out[base + (int64_t)c * T + s0 + s] = acc;
    }
  }
}

// Backward.  Per column block: dw = cont^T g, softmax backward over t, then the four gradients; dk / dcont accumulate
// over the column blocks in LDS and are written once at the end.
template <int LS_SB>
__global__ __launch_bounds__(256) void localstate_bwd_kernel(const LsArgs a) {
  constexpr int LS_LD = LS_SB + 1, NP = 256 / LS_SB;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int T = a.T, ch = a.ch, tid = threadIdx.x;
  float* kk = lds;                       // [ch][T]   k, later overwritten by nothing (read-only)
  float* cc = kk + ch * T;               // [ch][T]   content
  float* sc = cc + ch * T;               // [T][LS_LD] w, then ds
  float* dws = sc + T * LS_LD;           // [T][LS_LD] dw
  float* qs = dws + T * LS_LD;           // [ch][LS_SB] q block
  float* gs = qs + ch * LS_SB;           // [ch][LS_SB] gout block
  float* red = gs + ch * LS_SB;          // [NP][LS_SB] = 256 floats
  const int b = blockIdx.x / a.heads, hd = blockIdx.x % a.heads;
  const int64_t base = ((int64_t)b * a.heads + hd) * ch * T;
  const int64_t dbase = ((int64_t)b * a.heads + hd) * a.nd * T;
  const float* wb = a.w + ((int64_t)b * a.heads + hd) * T * T;
  const float inv = 1.0f / sqrtf((float)ch), invd = 1.0f / sqrtf((float)a.nd);
  for (int i = tid; i < ch * T; i += 256) { kk[i] = a.k[base + i]; cc[i] = a.cont[base + i]; }
  // dk / dcont accumulators live in registers: element i = tid + 256 * j  (ch * T <= 12288 -> j < 48)
  float dk_acc[48], dc_acc[48];
#pragma unroll
  for (int j = 0; j < 48; ++j) { dk_acc[j] = 0.f; dc_acc[j] = 0.f; }
  for (int s0 = 0; s0 < T; s0 += LS_SB) {
    __syncthreads();
    for (int i = tid; i < ch * LS_SB; i += 256) {
      const int c = i / LS_SB, s = i % LS_SB;
      const bool ok = s0 + s < T;
      qs[i] = ok ? a.q[base + (int64_t)c * T + s0 + s] : 0.f;
      gs[i] = ok ? a.gout[base + (int64_t)c * T + s0 + s] : 0.f;
    }
    for (int i = tid; i < T * LS_SB; i += 256) {
      const int t = i / LS_SB, s = i % LS_SB;
      sc[t * LS_LD + s] = s0 + s < T ? wb[(int64_t)t * T + s0 + s] : 0.f;
    }
    __syncthreads();
    // dw[t][s] = sum_c cont[c][t] g[c][s]
    for (int it = tid; it < T * (LS_SB / 4); it += 256) {
      const int t = it % T, s4 = (it / T) * 4;
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      for (int c = 0; c < ch; ++c) {
        const float cv = cc[c * T + t];
        const f32x4 gv = *reinterpret_cast<const f32x4*>(gs + c * LS_SB + s4);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = fmaf(cv, gv[j], acc[j]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) dws[t * LS_LD + s4 + j] = acc[j];
    }
    __syncthreads();
    // dcont[c][t] += sum_s w[t][s] g[c][s]   (uses w, before it is overwritten by ds)
#pragma unroll
    for (int j = 0; j < 48; ++j) {
      const int i = tid + 256 * j;
      if (i < ch * T) {
        const int c = i / T, t = i - c * T;
        float acc = 0.f;
#pragma unroll 8
        for (int s = 0; s < LS_SB; ++s) acc = fmaf(sc[t * LS_LD + s], gs[c * LS_SB + s], acc);
        dc_acc[j] += acc;
      }
    }
    __syncthreads();
    // softmax backward per column: ds = w * (dw - sum_t w dw); the masked diagonal carries no gradient
    {
      const int s = tid % LS_SB, p = tid / LS_SB;
      const int rows = (T + NP - 1) / NP, t0 = p * rows, t1 = min(t0 + rows, T);
      float dot = 0.f;
      for (int t = t0; t < t1; ++t) dot = fmaf(sc[t * LS_LD + s], dws[t * LS_LD + s], dot);
      red[p * LS_SB + s] = dot;
      __syncthreads();
      dot = 0.f;
#pragma unroll
      for (int i = 0; i < NP; ++i) dot += red[i * LS_SB + s];
      float dsum = 0.f;                              // sum_t ds[t][s] * |t - s|  (decay gradient)
      for (int t = t0; t < t1; ++t) {
        float ds = sc[t * LS_LD + s] * (dws[t * LS_LD + s] - dot);
        if (t == s0 + s) ds = 0.f;
        sc[t * LS_LD + s] = ds;
        dsum = fmaf(ds, fabsf((float)(t - (s0 + s))), dsum);
      }
      __syncthreads();
      red[p * LS_SB + s] = dsum;
      __syncthreads();
      if (p == 0 && s0 + s < T) {
        dsum = 0.f;
#pragma unroll
        for (int i = 0; i < NP; ++i) dsum += red[i * LS_SB + s];
        for (int f = 0; f < a.nd; ++f) {
          // d dots / d raw = -(f+1) |t-s| / sqrt(nd) * 0.5 * sig * (1 - sig)
          const float sg = ls_sigmoid(a.qd[dbase + (int64_t)f * T + s0 + s]);
          a.dqd[dbase + (int64_t)f * T + s0 + s] = -(float)(f + 1) * invd * 0.5f * sg * (1.f - sg) * dsum;
        }
      }
    }
    __syncthreads();
    // dq[c][s] = sum_t ds[t][s] k[c][t] / sqrt(ch)
    for (int i = tid; i < ch * LS_SB; i += 256) {
      const int c = i / LS_SB, s = i % LS_SB;
      float acc = 0.f;
      for (int t = 0; t < T; ++t) acc = fmaf(sc[t * LS_LD + s], kk[c * T + t], acc);
      if (s0 + s < T) a.dq[base + (int64_t)c * T + s0 + s] = acc * inv;
    }
    // dk[c][t] += sum_s ds[t][s] q[c][s] / sqrt(ch)
#pragma unroll
    for (int j = 0; j < 48; ++j) {
      const int i = tid + 256 * j;
      if (i < ch * T) {
        const int c = i / T, t = i - c * T;
        float acc = 0.f;
#pragma unroll 8
        for (int s = 0; s < LS_SB; ++s) acc = fmaf(sc[t * LS_LD + s], qs[c * LS_SB + s], acc);
        dk_acc[j] += acc * inv;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 48; ++j) {
    const int i = tid + 256 * j;
    if (i < ch * T) { a.dk[base + i] = dk_acc[j]; a.dcont[base + i] = dc_acc[j]; }
  }
}

static bool ls_ok(int B, int heads, int ch, int T, int nd) {
  return B > 0 && heads > 0 && ch > 0 && T > 0 && T <= 256 && nd > 0 && nd <= 8 && (int64_t)ch * T <= 12288;
}

extern "C" int rfx_localstate_fwd(const float* q, const float* k, const float* cont, const float* qd, int32_t B,
                                  int32_t heads, int32_t ch, int32_t T, int32_t nd, float* w, float* out, void* stream) {
  if (!q || !k || !cont || !qd || !out || !ls_ok(B, heads, ch, T, nd)) return -1;
  LsArgs a{};
  a.q = q; a.k = k; a.cont = cont; a.qd = qd; a.w = w; a.out = out;
  a.heads = heads; a.ch = ch; a.T = T; a.nd = nd;
  constexpr int SB = 32;
  const size_t lds = sizeof(float) * ((size_t)2 * ch * T + (size_t)T * (SB + 1) + (size_t)ch * SB + (size_t)nd * SB + 512);
  if (lds > 160 * 1024) return -1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(localstate_fwd_kernel<SB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds) != hipSuccess) return -3;
  hipLaunchKernelGGL(localstate_fwd_kernel<SB>, dim3(B * heads), dim3(256), lds, (hipStream_t)stream, a);
  RFX_CHECK_LAUNCH();
  return 0;
}

extern "C" int rfx_localstate_bwd(const float* q, const float* k, const float* cont, const float* qd, const float* w,
                                  const float* gout, int32_t B, int32_t heads, int32_t ch, int32_t T, int32_t nd, float* dq,
                                  float* dk, float* dcont, float* dqd, void* stream) {
  if (!q || !k || !cont || !qd || !w || !gout || !dq || !dk || !dcont || !dqd || !ls_ok(B, heads, ch, T, nd)) return -1;
  LsArgs a{};
  a.q = q; a.k = k; a.cont = cont; a.qd = qd; a.w = const_cast<float*>(w); a.gout = gout;
  a.dq = dq; a.dk = dk; a.dcont = dcont; a.dqd = dqd;
  a.heads = heads; a.ch = ch; a.T = T; a.nd = nd;
  constexpr int SB = 16;
  const size_t lds = sizeof(float) * ((size_t)2 * ch * T + (size_t)2 * T * (SB + 1) + (size_t)2 * ch * SB + 256);
  if (lds > 160 * 1024) return -1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(localstate_bwd_kernel<SB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds) != hipSuccess) return -3;
  hipLaunchKernelGGL(localstate_bwd_kernel<SB>, dim3(B * heads), dim3(256), lds, (hipStream_t)stream, a);
  RFX_CHECK_LAUNCH();
  return 0;
}

// ---- any-T path (streams the keys; no T x T block anywhere) --------------------------------------------------------
// The two kernels above hold k / content of a whole (sample, head) in LDS and therefore stop at T = 256 frames, i.e. one
// 262144-sample clip of the reference's configs.  Whole files (InferenceDataset, scripts/remfx_detect.py) are longer:
// this path handles any T with O(T) memory.  One wave per 64 query columns (forward, query-major backward) or 64 key rows
// (key-major backward); keys / queries stream through LDS in chunks of 64, lane-private vectors (q, accumulators) live in
// LDS columns [c][lane] (bank = lane: conflict-free).  Two passes over the keys in the forward (max + sum, then the
// weighted content sum) instead of a running rescale.  Exact fp32 on the vector ALU; this is the robustness path, not
// the fast one (T <= 256 keeps the kernels above / attention_mfma.hip).
// stat: (B*heads*T, 4) = {max, sum, delta = <out, gout> (written by the query-major backward), unused}
struct LsGenArgs {
  const float *q, *k, *cont, *qd, *out, *gout;
  float *stat, *o, *dq, *dk, *dcont, *dqd;
  int heads, ch, T, nd;
  int diag;            // 1: LocalState (diagonal masked to -100, decay penalty); 0: plain scaled-dot-product attention (nd = 0)
};

__device__ __forceinline__ float lsg_score(const float* rowv, const float* colv, int ch, float inv, float pen, int t, int s, int diag) {
  // rowv[c * 64] broadcast operand, colv[c * 64] lane operand
  float acc = 0.f;
  for (int c = 0; c < ch; ++c) acc = fmaf(rowv[c * 64], colv[c * 64], acc);
  const float v = acc * inv - fabsf((float)(t - s)) * pen;
  return (diag && t == s) ? -100.0f : v;
}

// mode 0: forward (stat{max,sum}, out); mode 1: query-major backward (dq, dqd, stat.delta)
template <int MODE>
__global__ __launch_bounds__(64) void localstate_gen_q_kernel(const LsGenArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int T = a.T, ch = a.ch, nd = a.nd, lane = threadIdx.x;
  float* qs = lds;                     // [ch][64] my queries
  float* ks = qs + ch * 64;            // [ch][64] key chunk
  float* cs = ks + ch * 64;            // [ch][64] content chunk
  float* acc = cs + ch * 64;           // [ch][64] out (fwd) / dq (bwd) accumulators
  float* gs = acc + ch * 64;           // [ch][64] gout tile (bwd only)
  const int bh = blockIdx.x, s = blockIdx.y * 64 + lane;
  const bool sv = s < T;
  const int64_t base = (int64_t)bh * ch * T, dbase = (int64_t)bh * nd * T;
  const float inv = 1.0f / sqrtf((float)ch), invd = 1.0f / sqrtf((float)nd);
  float pen = 0.f;                     // sum_f (f + 1) / sqrt(nd) * sigmoid(qd[f, s]) / 2
  for (int f = 0; f < nd; ++f) pen += sv ? (float)(f + 1) * invd * 0.5f * ls_sigmoid(a.qd[dbase + (int64_t)f * T + s]) : 0.f;
  for (int c = 0; c < ch; ++c) {
    qs[c * 64 + lane] = sv ? a.q[base + (int64_t)c * T + s] : 0.f;
    acc[c * 64 + lane] = 0.f;
    if (MODE == 1) gs[c * 64 + lane] = sv ? a.gout[base + (int64_t)c * T + s] : 0.f;
  }
  float m = -3.0e38f, l = 0.f, delta = 0.f, A = 0.f;
  if (MODE == 1) {
    if (sv) { m = a.stat[((int64_t)bh * T + s) * 4]; l = a.stat[((int64_t)bh * T + s) * 4 + 1]; }
    for (int c = 0; c < ch; ++c) delta = fmaf(sv ? a.out[base + (int64_t)c * T + s] : 0.f, gs[c * 64 + lane], delta);
    if (sv) a.stat[((int64_t)bh * T + s) * 4 + 2] = delta;
  }
  const int npass = MODE == 0 ? 2 : 1;
  for (int pass = 0; pass < npass; ++pass) {
    const bool second = MODE == 1 || pass == 1;
    const float rl = second ? 1.0f / l : 0.f;
    for (int t0 = 0; t0 < T; t0 += 64) {
      __syncthreads();
      const bool tv = t0 + lane < T;
      for (int c = 0; c < ch; ++c) {
        ks[c * 64 + lane] = tv ? a.k[base + (int64_t)c * T + t0 + lane] : 0.f;
        if (second) cs[c * 64 + lane] = tv ? a.cont[base + (int64_t)c * T + t0 + lane] : 0.f;
      }
      __syncthreads();
      const int nj = min(64, T - t0);
      for (int j = 0; j < nj; ++j) {
        const int t = t0 + j;
        const float v = lsg_score(ks + j, qs + lane, ch, inv, pen, t, s, a.diag);
        if (!second) {
          const float mn = fmaxf(m, v);
          l = l * expf(m - mn) + expf(v - mn);
          m = mn;
        } else {
          const float p = expf(v - m) * rl;
          if (MODE == 0) {
            for (int c = 0; c < ch; ++c) acc[c * 64 + lane] = fmaf(p, cs[c * 64 + j], acc[c * 64 + lane]);
          } else {
            float dw = 0.f;
            for (int c = 0; c < ch; ++c) dw = fmaf(cs[c * 64 + j], gs[c * 64 + lane], dw);
            const float dv = (a.diag && t == s) ? 0.f : p * (dw - delta);          // the masked diagonal is a constant
            A = fmaf(dv, fabsf((float)(t - s)), A);
            const float dvi = dv * inv;
            for (int c = 0; c < ch; ++c) acc[c * 64 + lane] = fmaf(dvi, ks[c * 64 + j], acc[c * 64 + lane]);
          }
        }
      }
    }
  }
  if (!sv) return;
  if (MODE == 0) {
    a.stat[((int64_t)bh * T + s) * 4] = m;
    a.stat[((int64_t)bh * T + s) * 4 + 1] = l;
    for (int c = 0; c < ch; ++c) a.o[base + (int64_t)c * T + s] = acc[c * 64 + lane];
  } else {
    for (int c = 0; c < ch; ++c) a.dq[base + (int64_t)c * T + s] = acc[c * 64 + lane];
    for (int f = 0; f < nd; ++f) {
      const float sg = ls_sigmoid(a.qd[dbase + (int64_t)f * T + s]);
      a.dqd[dbase + (int64_t)f * T + s] = -(float)(f + 1) * invd * A * 0.5f * sg * (1.0f - sg);
    }
  }
}

// key-major backward: lane = key row t; dk[c][t], dcont[c][t]
__global__ __launch_bounds__(64) void localstate_gen_k_kernel(const LsGenArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int T = a.T, ch = a.ch, nd = a.nd, lane = threadIdx.x;
  float* km = lds;                     // [ch][64] my keys
  float* cm = km + ch * 64;            // [ch][64] my content rows
  float* dka = cm + ch * 64;           // [ch][64]
  float* dca = dka + ch * 64;          // [ch][64]
  float* qc = dca + ch * 64;           // [ch][64] query chunk
  float* gc = qc + ch * 64;            // [ch][64] gout chunk
  float* sc = gc + ch * 64;            // [4][64] per query of the chunk: max, 1 / sum, delta, pen
  const int bh = blockIdx.x, t = blockIdx.y * 64 + lane;
  const bool tv = t < T;
  const int64_t base = (int64_t)bh * ch * T, dbase = (int64_t)bh * nd * T;
  const float inv = 1.0f / sqrtf((float)ch), invd = 1.0f / sqrtf((float)nd);
  for (int c = 0; c < ch; ++c) {
    km[c * 64 + lane] = tv ? a.k[base + (int64_t)c * T + t] : 0.f;
    cm[c * 64 + lane] = tv ? a.cont[base + (int64_t)c * T + t] : 0.f;
    dka[c * 64 + lane] = 0.f;
    dca[c * 64 + lane] = 0.f;
  }
  for (int s0 = 0; s0 < T; s0 += 64) {
    __syncthreads();
    const int s = s0 + lane;
    const bool sv = s < T;
    for (int c = 0; c < ch; ++c) {
      qc[c * 64 + lane] = sv ? a.q[base + (int64_t)c * T + s] : 0.f;
      gc[c * 64 + lane] = sv ? a.gout[base + (int64_t)c * T + s] : 0.f;
    }
    float pen = 0.f;
    for (int f = 0; f < nd; ++f) pen += sv ? (float)(f + 1) * invd * 0.5f * ls_sigmoid(a.qd[dbase + (int64_t)f * T + s]) : 0.f;
    sc[lane] = sv ? a.stat[((int64_t)bh * T + s) * 4] : 0.f;
    sc[64 + lane] = sv ? 1.0f / a.stat[((int64_t)bh * T + s) * 4 + 1] : 0.f;
    sc[128 + lane] = sv ? a.stat[((int64_t)bh * T + s) * 4 + 2] : 0.f;
    sc[192 + lane] = pen;
    __syncthreads();
    const int nj = min(64, T - s0);
    for (int j = 0; j < nj; ++j) {
      const int sj = s0 + j;
      const float v = lsg_score(qc + j, km + lane, ch, inv, sc[192 + j], t, sj, a.diag);
      const float p = expf(v - sc[j]) * sc[64 + j];
      float dw = 0.f;
      for (int c = 0; c < ch; ++c) dw = fmaf(cm[c * 64 + lane], gc[c * 64 + j], dw);
      const float dvi = ((a.diag && t == sj) ? 0.f : p * (dw - sc[128 + j])) * inv;
      for (int c = 0; c < ch; ++c) {
        dca[c * 64 + lane] = fmaf(p, gc[c * 64 + j], dca[c * 64 + lane]);
        dka[c * 64 + lane] = fmaf(dvi, qc[c * 64 + j], dka[c * 64 + lane]);
      }
    }
  }
  if (!tv) return;
  for (int c = 0; c < ch; ++c) {
    a.dk[base + (int64_t)c * T + t] = dka[c * 64 + lane];
    a.dcont[base + (int64_t)c * T + t] = dca[c * 64 + lane];
  }
}

static bool lsg_ok(int B, int heads, int ch, int T, int nd) {
  return B > 0 && heads > 0 && ch > 0 && ch <= 104 && T > 0 && nd >= 0 && nd <= 64 &&
         (int64_t)B * heads <= 0x7fffffff && (T + 63) / 64 <= 65535;
}

template <typename K>
static int lsg_launch(K kern, const LsGenArgs& a, int B, size_t lds, void* stream) {
  if (lds > 160 * 1024) return -1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return -3;
  hipLaunchKernelGGL(kern, dim3(B * a.heads, (a.T + 63) / 64), dim3(64), lds, (hipStream_t)stream, a);
  RFX_CHECK_LAUNCH();
  return 0;
}

extern "C" int rfx_localstate_gen_fwd(const float* q, const float* k, const float* cont, const float* qd, int32_t B,
                                      int32_t heads, int32_t ch, int32_t T, int32_t nd, float* stat, float* out, void* stream) {
  if (!q || !k || !cont || !qd || !stat || !out || nd <= 0 || !lsg_ok(B, heads, ch, T, nd)) return -1;
  LsGenArgs a{};
  a.q = q; a.k = k; a.cont = cont; a.qd = qd; a.stat = stat; a.o = out;
  a.heads = heads; a.ch = ch; a.T = T; a.nd = nd; a.diag = 1;
  return lsg_launch(localstate_gen_q_kernel<0>, a, B, sizeof(float) * 5 * ch * 64, stream);
}

extern "C" int rfx_localstate_gen_bwd(const float* q, const float* k, const float* cont, const float* qd, float* stat,
                                      const float* out, const float* gout, int32_t B, int32_t heads, int32_t ch, int32_t T,
                                      int32_t nd, float* dq, float* dk, float* dcont, float* dqd, void* stream) {
  if (!q || !k || !cont || !qd || !stat || !out || !gout || !dq || !dk || !dcont || !dqd || nd <= 0 || !lsg_ok(B, heads, ch, T, nd)) return -1;
  LsGenArgs a{};
  a.q = q; a.k = k; a.cont = cont; a.qd = qd; a.stat = stat; a.out = out; a.gout = gout;
  a.dq = dq; a.dk = dk; a.dcont = dcont; a.dqd = dqd;
  a.heads = heads; a.ch = ch; a.T = T; a.nd = nd; a.diag = 1;
  int rc = lsg_launch(localstate_gen_q_kernel<1>, a, B, sizeof(float) * 5 * ch * 64, stream);     // writes stat.delta first
  if (rc) return rc;
  return lsg_launch(localstate_gen_k_kernel, a, B, sizeof(float) * (6 * ch * 64 + 256), stream);
}

// Plain multi-head scaled-dot-product attention on the same streaming kernels (no decay penalty, no diagonal mask):
//   out[c, s] = sum_t softmax_t(<k[:, t], q[:, s]> / sqrt(ch)) v[c, t]        q, k, v: (B, heads * ch, T) channel-major
// nn.MultiheadAttention core of asteroid's DPTNet `ImprovedTransformedLayer` (reference remfx/models.py:327-344).
extern "C" int rfx_mha_fwd(const float* q, const float* k, const float* v, int32_t B, int32_t heads, int32_t ch, int32_t T, float* stat,
                           float* out, void* stream) {
  if (!q || !k || !v || !stat || !out || !lsg_ok(B, heads, ch, T, 0)) return -1;
  LsGenArgs a{};
  a.q = q; a.k = k; a.cont = v; a.qd = nullptr; a.stat = stat; a.o = out;
  a.heads = heads; a.ch = ch; a.T = T; a.nd = 0; a.diag = 0;
  return lsg_launch(localstate_gen_q_kernel<0>, a, B, sizeof(float) * 5 * ch * 64, stream);
}
extern "C" int rfx_mha_bwd(const float* q, const float* k, const float* v, float* stat, const float* out, const float* gout, int32_t B,
                           int32_t heads, int32_t ch, int32_t T, float* dq, float* dk, float* dv, void* stream) {
  if (!q || !k || !v || !stat || !out || !gout || !dq || !dk || !dv || !lsg_ok(B, heads, ch, T, 0)) return -1;
  LsGenArgs a{};
  a.q = q; a.k = k; a.cont = v; a.qd = nullptr; a.stat = stat; a.out = out; a.gout = gout;
  a.dq = dq; a.dk = dk; a.dcont = dv; a.dqd = nullptr;
  a.heads = heads; a.ch = ch; a.T = T; a.nd = 0; a.diag = 0;
  int rc = lsg_launch(localstate_gen_q_kernel<1>, a, B, sizeof(float) * 5 * ch * 64, stream);
  if (rc) return rc;
  return lsg_launch(localstate_gen_k_kernel, a, B, sizeof(float) * (6 * ch * 64 + 256), stream);
}
