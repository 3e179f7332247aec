// LocalState attention on the bf16 matrix pipe (bf16 mode only; the exact fp32 kernels of attention.hip stay the f32 / bf16x3
// path and the fallback for head widths that are not a multiple of 16).  Same operator as attention.hip:
//   dots[t, s] = <k[:, t], q[:, s]> / sqrt(ch) - |t - s| * D[s],  D[s] = sum_f (f + 1) sigmoid(qd[f, s]) / (2 sqrt(nd)),
//   dots[s, s] = -100,  w = softmax over t,  out[c, s] = sum_t cont[c, t] w[t, s]
// (torchaudio HDemucs `_LocalState`, reached from remfx/models.py:319).  Under torch autocast -- what trainer.precision=bf16-mixed
// means in the reference -- both einsums run on bf16 operands with fp32 accumulation and the softmax in fp32: that is what this
// file does.  r02 profile: the fp32 VALU kernels ran one workgroup per CU at 11 TF/s (2.2 + 3.6 ms per Demucs step).
//
// Flash-style, v_mfma_f32_32x32x16_bf16.  A wave owns 32 query columns s; the whole score block S[0..T) x 32 lives in its
// accumulators (8 tiles x 16 registers), so the softmax over t is a reduction over a lane's registers plus one cross-half
// shuffle.  The 32x32 C/D layout (lane (j, h): rows (r&3) + 8 (r>>2) + 4h of column j) IS a valid B operand of the next
// MFMA if its A operand enumerates k in the same permuted order: K step (R, u) of the product content . P covers
// t = 32R + 16u + 4h + (e & 3) + 8 (e >> 2), e = 0..7, i.e. two runs of four consecutive t -- two ds_read_b64 from a natural
// [channel][t] bf16 tile in LDS.  P never leaves registers and is never written to memory: the backward pass recomputes it
// (no (B, heads, T, T) tensor).  Operands that are the A matrix with k = channel (k^T, content^T, q^T, gout^T) are read straight
// from global memory: a lane needs 8 channels of one position, i.e. 8 loads that are coalesced across the lanes.
#include "common.h"

typedef __bf16 lm_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 lm_bf16x2 __attribute__((ext_vector_type(2)));
typedef float lm_f32x2 __attribute__((ext_vector_type(2)));

struct LmArgs {
  const float *q, *k, *cont, *qd, *gout;   // (B, heads*ch, T) x3, (B, heads*nd, T), d out
  float *out, *dq, *dk, *dcont, *dqd;
  float4* stat;                            // backward workspace: (max, 1/sum, delta, D) per (batch row, head, query column)
  int T, nd;
};

constexpr int LM_TP = 264;                 // row stride (bf16 elements) of the natural [channel][t] tiles in LDS

__device__ __forceinline__ uint32_t lm_pk(float a, float b) {
  const lm_f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, lm_bf16x2));
}
__device__ __forceinline__ lm_bf16x8 lm_pack8(const float (&x)[8]) {
  return __builtin_bit_cast(lm_bf16x8, make_uint4(lm_pk(x[0], x[1]), lm_pk(x[2], x[3]), lm_pk(x[4], x[5]), lm_pk(x[6], x[7])));
}
// 8 consecutive channels of one position: p[e * T], e = 0..7 (p already points at a valid, clamped position)
__device__ __forceinline__ lm_bf16x8 lm_ldfrag(const float* __restrict__ p, int T, bool ok) {
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = p[(int64_t)e * T];
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = ok ? x[e] : 0.f;
  return lm_pack8(x);
}
// A fragment of a natural [channel][t] tile for K step (R, u): rows c = 32 ct + l31, k = the permuted t order above
__device__ __forceinline__ lm_bf16x8 lm_nat_frag(const uint16_t* tile, int c, int t0) {
  const uint16_t* rp = tile + c * LM_TP + t0;
  const uint2 lo = *reinterpret_cast<const uint2*>(rp);
  const uint2 hi = *reinterpret_cast<const uint2*>(rp + 8);
  return __builtin_bit_cast(lm_bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
}
// natural tile fill: rows >= CH and positions >= T are zero; thread = position
template <int CH, int CT>
__device__ __forceinline__ void lm_fill_nat(uint16_t* tile, const float* __restrict__ src, int T, int tid) {
  const bool tv = tid < T;
  const float* sp = src + (tv ? tid : 0);
#pragma unroll 8
  for (int c = 0; c < CT * 32; ++c) {
    const float v = (c < CH && tv) ? sp[(int64_t)(c < CH ? c : 0) * T] : 0.f;
    tile[c * LM_TP + tid] = (uint16_t)(lm_pk(v, 0.f) & 0xffffu);
  }
}
__device__ __forceinline__ float lm_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ int lm_row(int R, int r, int h) { return 32 * R + (r & 3) + 8 * (r >> 2) + 4 * h; }

// scores of 32 query columns against all T keys + softmax over t, left normalised in acc; returns the column's max and 1 / sum
template <int KS>
__device__ __forceinline__ void lm_scores_softmax(const float* __restrict__ kbase, int T, int s, int l31, int h, float D, float inv,
                                                  const lm_bf16x8 (&qf)[KS], f32x16 (&acc)[8], float& m_out, float& rinv_out) {
#pragma unroll
  for (int R = 0; R < 8; ++R) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[R][r] = 0.f;
    if (32 * R < T) {                                   // wave-uniform
      asm volatile("" ::: "memory");                    // keep the fragment loads of row tile R + 1 behind the MFMAs of tile R
      const int t = 32 * R + l31;
      const bool tv = t < T;
      const float* kp = kbase + (tv ? t : T - 1);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const lm_bf16x8 af = lm_ldfrag(kp + (int64_t)(16 * ks + 8 * h) * T, T, tv);
        acc[R] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, qf[ks], acc[R], 0, 0, 0);
      }
    }
  }
  float m = -3.0e38f;
#pragma unroll
  for (int R = 0; R < 8; ++R)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = lm_row(R, r, h);
      float v = acc[R][r] * inv - fabsf((float)(t - s)) * D;
      v = (t == s) ? -100.0f : v;
      v = t < T ? v : -3.0e38f;
      acc[R][r] = v;
      m = fmaxf(m, v);
    }
  m = fmaxf(m, __shfl_xor(m, 32));
  float sum = 0.f;
#pragma unroll
  for (int R = 0; R < 8; ++R)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __expf(acc[R][r] - m);
      acc[R][r] = e;
      sum += e;
    }
  sum += __shfl_xor(sum, 32);
  const float rinv = 1.0f / sum;
#pragma unroll
  for (int R = 0; R < 8; ++R)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[R][r] *= rinv;
  m_out = m; rinv_out = rinv;
}

template <int KS>
__global__ __launch_bounds__(256) void ls_mfma_fwd_kernel(const LmArgs a) {
  constexpr int CH = 16 * KS, CT = (CH + 31) / 32;
  extern __shared__ __attribute__((aligned(16))) uint16_t lm_lds[];
  uint16_t* cn = lm_lds;                                 // content, natural [CT*32][LM_TP]
  const int T = a.T, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int64_t base = (int64_t)blockIdx.x * CH * T, dbase = (int64_t)blockIdx.x * a.nd * T;
  lm_fill_nat<CH, CT>(cn, a.cont + base, T, tid);
  __syncthreads();
  const int s = blockIdx.y * 128 + wave * 32 + l31;
  if (blockIdx.y * 128 + wave * 32 >= T) return;         // wave-uniform
  const bool sv = s < T;
  const int scl = sv ? s : T - 1;
  const float inv = 1.0f / sqrtf((float)CH);
  lm_bf16x8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qf[ks] = lm_ldfrag(a.q + base + (int64_t)(16 * ks + 8 * h) * T + scl, T, sv);
  float D = 0.f;
  for (int f = 0; f < a.nd; ++f) D += (float)(f + 1) * lm_sigmoid(a.qd[dbase + (int64_t)f * T + scl]);
  D *= 0.5f / sqrtf((float)a.nd);
  f32x16 acc[8];
  float m, rinv;
  lm_scores_softmax<KS>(a.k + base, T, s, l31, h, D, inv, qf, acc, m, rinv);
  f32x16 o[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[ct][r] = 0.f;
#pragma unroll
  for (int R = 0; R < 8; ++R) {
    if (32 * R >= T) continue;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float pv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) pv[e] = acc[R][8 * u + e];
      const lm_bf16x8 pf = lm_pack8(pv);
      const int t0 = 32 * R + 16 * u + 4 * h;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
        o[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lm_nat_frag(cn, 32 * ct + l31, t0), pf, o[ct], 0, 0, 0);
    }
  }
  if (!sv) return;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = 32 * ct + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (c < CH) a.out[base + (int64_t)c * T + s] = o[ct][r];
    }
}

// Backward in two launches of 4-wave workgroups, 128 columns each.
// Pass A (lane = query column s): recompute P, delta[s] = sum_t P dP, then dS = P (dP - delta) feeds dq (A = k natural) and the decay
// gradient; the per-column statistics (max, 1/sum, delta, D) go to `stat` (B*heads*T float4, caller-owned).
// Pass B (lane = key position t): recompute P^T / dS^T tile by tile from those statistics; dk = q . dS^T, dcont = gout . P^T
// (A = q / gout natural).
template <int KS>
__global__ __launch_bounds__(256) void ls_mfma_bwd_a_kernel(const LmArgs a) {
  constexpr int CH = 16 * KS, CT = (CH + 31) / 32;
  extern __shared__ __attribute__((aligned(16))) uint16_t lm_lds[];
  uint16_t* kn = lm_lds;                                 // k natural
  const int T = a.T, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int64_t base = (int64_t)blockIdx.x * CH * T, dbase = (int64_t)blockIdx.x * a.nd * T;
  const float inv = 1.0f / sqrtf((float)CH), invd = 1.0f / sqrtf((float)a.nd);
  lm_fill_nat<CH, CT>(kn, a.k + base, T, tid);
  __syncthreads();
  if (blockIdx.y * 128 + wave * 32 >= T) return;         // wave-uniform
  const int s = blockIdx.y * 128 + wave * 32 + l31;
  const bool sv = s < T;
  const int scl = sv ? s : T - 1;
  lm_bf16x8 qf[KS], gf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    qf[ks] = lm_ldfrag(a.q + base + (int64_t)(16 * ks + 8 * h) * T + scl, T, sv);
    gf[ks] = lm_ldfrag(a.gout + base + (int64_t)(16 * ks + 8 * h) * T + scl, T, sv);
  }
  float D = 0.f;
  for (int f = 0; f < a.nd; ++f) D += (float)(f + 1) * lm_sigmoid(a.qd[dbase + (int64_t)f * T + scl]);
  D *= 0.5f * invd;
  f32x16 acc[8];
  float m, rinv;
  lm_scores_softmax<KS>(a.k + base, T, s, l31, h, D, inv, qf, acc, m, rinv);
  // sweep 1: delta = sum_t P[t, s] dP[t, s],  dP[t, s] = sum_c cont[c, t] gout[c, s]
  float delta = 0.f;
#pragma unroll
  for (int R = 0; R < 8; ++R) {
    if (32 * R >= T) continue;
    // the R loop must be unrolled (acc[R] lives in registers); without a barrier hipcc hoists the fragment loads of all row
    // tiles to the top and spills
    asm volatile("" ::: "memory");
    const int t = 32 * R + l31;
    const bool tv = t < T;
    const float* cp = a.cont + base + (tv ? t : T - 1);
    f32x16 dP;
#pragma unroll
    for (int r = 0; r < 16; ++r) dP[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lm_ldfrag(cp + (int64_t)(16 * ks + 8 * h) * T, T, tv), gf[ks], dP, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) delta = fmaf(acc[R][r], dP[r], delta);
  }
  delta += __shfl_xor(delta, 32);
  if (h == 0 && sv) a.stat[(int64_t)blockIdx.x * T + s] = make_float4(m, rinv, delta, D);
  // sweep 2: dS, dq, decay gradient
  f32x16 dq[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[ct][r] = 0.f;
  float dsum = 0.f;
#pragma unroll
  for (int R = 0; R < 8; ++R) {
    if (32 * R >= T) continue;
    asm volatile("" ::: "memory");
    const int t = 32 * R + l31;
    const bool tv = t < T;
    const float* cp = a.cont + base + (tv ? t : T - 1);
    f32x16 dP;
#pragma unroll
    for (int r = 0; r < 16; ++r) dP[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lm_ldfrag(cp + (int64_t)(16 * ks + 8 * h) * T, T, tv), gf[ks], dP, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float pv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int r = 8 * u + e;
        const int tr = lm_row(R, r, h);
        float ds = acc[R][r] * (dP[r] - delta);
        ds = (tr == s) ? 0.f : ds;                       // the masked diagonal is a constant
        pv[e] = ds;
        dsum = fmaf(ds, fabsf((float)(tr - s)), dsum);
      }
      const lm_bf16x8 pf = lm_pack8(pv);
      const int t0 = 32 * R + 16 * u + 4 * h;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
        dq[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lm_nat_frag(kn, 32 * ct + l31, t0), pf, dq[ct], 0, 0, 0);
    }
  }
  dsum += __shfl_xor(dsum, 32);
  if (!sv) return;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = 32 * ct + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (c < CH) a.dq[base + (int64_t)c * T + s] = dq[ct][r] * inv;
    }
  if (h == 0)
    for (int f = 0; f < a.nd; ++f) {
      const float sg = lm_sigmoid(a.qd[dbase + (int64_t)f * T + s]);
      a.dqd[dbase + (int64_t)f * T + s] = -(float)(f + 1) * invd * 0.5f * sg * (1.f - sg) * dsum;
    }
}

template <int KS>
__global__ __launch_bounds__(256) void ls_mfma_bwd_b_kernel(const LmArgs a) {
  constexpr int CH = 16 * KS, CT = (CH + 31) / 32, TILE = CT * 32 * LM_TP;
  extern __shared__ __attribute__((aligned(16))) uint16_t lm_lds[];
  uint16_t* qn = lm_lds;                                 // q natural
  uint16_t* gn = lm_lds + TILE;                          // gout natural
  float4* stat = reinterpret_cast<float4*>(lm_lds + 2 * TILE);      // [256] (max, 1/sum, delta, D) per query column
  const int T = a.T, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int64_t base = (int64_t)blockIdx.x * CH * T;
  const float inv = 1.0f / sqrtf((float)CH);
  lm_fill_nat<CH, CT>(qn, a.q + base, T, tid);
  lm_fill_nat<CH, CT>(gn, a.gout + base, T, tid);
  stat[tid] = tid < T ? a.stat[(int64_t)blockIdx.x * T + tid] : make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  if (blockIdx.y * 128 + wave * 32 >= T) return;         // wave-uniform
  const int t = blockIdx.y * 128 + wave * 32 + l31;
  const bool tv = t < T;
  const int tcl = tv ? t : T - 1;
  lm_bf16x8 kf[KS], cf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    kf[ks] = lm_ldfrag(a.k + base + (int64_t)(16 * ks + 8 * h) * T + tcl, T, tv);
    cf[ks] = lm_ldfrag(a.cont + base + (int64_t)(16 * ks + 8 * h) * T + tcl, T, tv);
  }
  f32x16 dK[CT], dC[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dK[ct][r] = 0.f; dC[ct][r] = 0.f; }
#pragma unroll 1
  for (int R = 0; R < 8; ++R) {
    if (32 * R >= T) break;
    const int sr = 32 * R + l31;
    const bool srv = sr < T;
    const int64_t soff = base + (srv ? sr : T - 1);
    f32x16 St, dPt;
#pragma unroll
    for (int r = 0; r < 16; ++r) { St[r] = 0.f; dPt[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      St = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lm_ldfrag(a.q + soff + (int64_t)(16 * ks + 8 * h) * T, T, srv), kf[ks], St, 0, 0, 0);
      dPt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lm_ldfrag(a.gout + soff + (int64_t)(16 * ks + 8 * h) * T, T, srv), cf[ks], dPt, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float p8[8], d8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int r = 8 * u + e;
        const int srow = lm_row(R, r, h);
        const float4 st = stat[srow & 255];
        float v = St[r] * inv - fabsf((float)(t - srow)) * st.w;
        v = (t == srow) ? -100.0f : v;
        const float p = (srow < T && tv) ? __expf(v - st.x) * st.y : 0.f;
        float ds = p * (dPt[r] - st.z);
        ds = (t == srow) ? 0.f : ds;
        p8[e] = p; d8[e] = ds;
      }
      const lm_bf16x8 pf = lm_pack8(p8), df = lm_pack8(d8);
      const int s0 = 32 * R + 16 * u + 4 * h;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        dK[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lm_nat_frag(qn, 32 * ct + l31, s0), df, dK[ct], 0, 0, 0);
        dC[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lm_nat_frag(gn, 32 * ct + l31, s0), pf, dC[ct], 0, 0, 0);
      }
    }
  }
  if (!tv) return;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = 32 * ct + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (c < CH) {
        a.dk[base + (int64_t)c * T + t] = dK[ct][r] * inv;
        a.dcont[base + (int64_t)c * T + t] = dC[ct][r];
      }
    }
}

static bool lm_ok(int B, int heads, int ch, int T, int nd) {
  return B > 0 && heads > 0 && T > 0 && T <= 256 && nd > 0 && nd <= 8 && ch % 16 == 0 && ch >= 16 && ch <= 96 && ch != 80;
}

template <int KS>
static int lm_launch_fwd(const LmArgs& a, int BH, hipStream_t s) {
  constexpr int CT = (16 * KS + 31) / 32;
  const size_t lds = sizeof(uint16_t) * CT * 32 * LM_TP;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(ls_mfma_fwd_kernel<KS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds) != hipSuccess) return -3;
  hipLaunchKernelGGL(ls_mfma_fwd_kernel<KS>, dim3(BH, (a.T + 127) / 128), dim3(256), lds, s, a);
  RFX_CHECK_LAUNCH();
  return 0;
}
template <int KS>
static int lm_launch_bwd(const LmArgs& a, int BH, hipStream_t s) {
  constexpr int CT = (16 * KS + 31) / 32;
  const size_t lds_a = sizeof(uint16_t) * CT * 32 * LM_TP;
  const size_t lds_b = sizeof(uint16_t) * 2 * CT * 32 * LM_TP + sizeof(float4) * 256;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(ls_mfma_bwd_a_kernel<KS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds_a) != hipSuccess) return -3;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(ls_mfma_bwd_b_kernel<KS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds_b) != hipSuccess) return -3;
  const dim3 grid(BH, (a.T + 127) / 128);
  hipLaunchKernelGGL(ls_mfma_bwd_a_kernel<KS>, grid, dim3(256), lds_a, s, a);
  RFX_CHECK_LAUNCH();
  hipLaunchKernelGGL(ls_mfma_bwd_b_kernel<KS>, grid, dim3(256), lds_b, s, a);
  RFX_CHECK_LAUNCH();
  return 0;
}

extern "C" int rfx_localstate_mfma_ok(int32_t B, int32_t heads, int32_t ch, int32_t T, int32_t nd) {
  return lm_ok(B, heads, ch, T, nd) ? 1 : 0;
}

extern "C" int rfx_localstate_mfma_fwd(const float* q, const float* k, const float* cont, const float* qd, int32_t B,
                                       int32_t heads, int32_t ch, int32_t T, int32_t nd, float* out, void* stream) {
  if (!q || !k || !cont || !qd || !out || !lm_ok(B, heads, ch, T, nd)) return -1;
  LmArgs a{};
  a.q = q; a.k = k; a.cont = cont; a.qd = qd; a.out = out; a.T = T; a.nd = nd;
  hipStream_t s = (hipStream_t)stream;
  switch (ch / 16) {
    case 1: return lm_launch_fwd<1>(a, B * heads, s);
    case 2: return lm_launch_fwd<2>(a, B * heads, s);
    case 3: return lm_launch_fwd<3>(a, B * heads, s);
    case 4: return lm_launch_fwd<4>(a, B * heads, s);
    default: return lm_launch_fwd<6>(a, B * heads, s);
  }
}

extern "C" int rfx_localstate_mfma_bwd(const float* q, const float* k, const float* cont, const float* qd, const float* gout,
                                       int32_t B, int32_t heads, int32_t ch, int32_t T, int32_t nd, float* dq, float* dk,
                                       float* dcont, float* dqd, float* stat, void* stream) {
  if (!q || !k || !cont || !qd || !gout || !dq || !dk || !dcont || !dqd || !stat || ((uintptr_t)stat & 15) ||
      !lm_ok(B, heads, ch, T, nd)) return -1;
  LmArgs a{};
  a.q = q; a.k = k; a.cont = cont; a.qd = qd; a.gout = gout; a.dq = dq; a.dk = dk; a.dcont = dcont; a.dqd = dqd;
  a.stat = reinterpret_cast<float4*>(stat);
  a.T = T; a.nd = nd;
  hipStream_t s = (hipStream_t)stream;
  switch (ch / 16) {
    case 1: return lm_launch_bwd<1>(a, B * heads, s);
    case 2: return lm_launch_bwd<2>(a, B * heads, s);
    case 3: return lm_launch_bwd<3>(a, B * heads, s);
    case 4: return lm_launch_bwd<4>(a, B * heads, s);
    default: return lm_launch_bwd<6>(a, B * heads, s);
  }
}
