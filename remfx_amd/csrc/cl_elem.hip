// Channels-last bf16 family: layout conversion at the ends of the channels-last trunk (the spectrogram branch of Hybrid Demucs
// enters from the STFT / the merged deep layers in channel-major fp32 and leaves the same way; torchaudio HDemucs behind
// remfx/models.py:308,317) and the small elementwise kernels of that trunk.
#include "cl_common.h"

// (N, C, A, B) channel-major, B contiguous  ->  [N][A][B][bs] channels-last bf16.  Tile = 64 positions x 32 channels through LDS.
// MODE: 0 store | 1 v * gelu'(aux) | 2 GLU backward against aux = stored [a | b] (dst carries 2 C channels: [ga | gb]) | 3 gelu(v)
// res (optional, any mode): v += res first.  v and the sum are rounded to bf16 where a stored tensor would have been (the fused
// passes replace "convert, store, re-read" chains and keep their roundings).
template <typename T, int MODE>
__global__ __launch_bounds__(256) void cl_from_cm_kernel(const T* __restrict__ src, int64_t s_ns, int64_t s_cs, int64_t s_as, int C,
                                                         int A, int B, rfx_cl_tensor dst, rfx_cl_tensor res, rfx_cl_tensor aux) {
  __shared__ uint16_t tile[32][66];
  const int tpb = B / 64;
  const int pt = blockIdx.x, cgp = blockIdx.y;
  const int n = pt / (A * tpb), rem = pt - n * A * tpb, a = rem / tpb, b0 = (rem - a * tpb) * 64;
  const int t = threadIdx.x, p = t & 63, cr = t >> 6;
  const T* s = src + (int64_t)n * s_ns + (int64_t)a * s_as + b0 + p;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = cr + 4 * k, cc = cgp * 32 + c;
    tile[c][p] = (uint16_t)rfx_bf16_bits(cc < C ? rfx_ld1(s + (int64_t)cc * s_cs) : 0.f);
  }
  __syncthreads();
  const int pp = t >> 2, g8 = t & 3, c0 = cgp * 32 + g8 * 8;
  if (c0 < C) {
    uint32_t w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = (uint32_t)tile[g8 * 8 + 2 * e][pp] | ((uint32_t)tile[g8 * 8 + 2 * e + 1][pp] << 16);
    uint4 raw = make_uint4(w[0], w[1], w[2], w[3]);
    auto at = [&](const rfx_cl_tensor& q, int ch) {
      return reinterpret_cast<uint16_t*>(q.p) + (int64_t)n * q.ns + (int64_t)a * q.as + (int64_t)(b0 + pp) * q.bs + q.c0 + ch;
    };
    float v[8];
    if (res.p != nullptr) {
      float r[8];
      cl_unpack8(raw, v);
      cl_unpack8(*reinterpret_cast<const uint4*>(at(res, c0)), r);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += r[e];
      raw = cl_pack8(v);
    }
    if (MODE == 0) {
      *reinterpret_cast<uint4*>(at(dst, c0)) = raw;
    } else if (MODE == 3) {
      float o[8];
      cl_unpack8(raw, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = rfx_gelu(v[e]);
      *reinterpret_cast<uint4*>(at(dst, c0)) = cl_pack8(o);
    } else if (MODE == 1) {
      float z[8], o[8];
      cl_unpack8(raw, v);
      cl_unpack8(*reinterpret_cast<const uint4*>(at(aux, c0)), z);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = v[e] * rfx_gelu_grad(z[e]);
      *reinterpret_cast<uint4*>(at(dst, c0)) = cl_pack8(o);
    } else {
      float fa[8], fb[8], ga[8], gb[8];
      cl_unpack8(raw, v);
      cl_unpack8(*reinterpret_cast<const uint4*>(at(aux, c0)), fa);
      cl_unpack8(*reinterpret_cast<const uint4*>(at(aux, C + c0)), fb);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float sg = rfx_sigmoid(fb[e]);
        ga[e] = v[e] * sg;
        gb[e] = v[e] * fa[e] * sg * (1.f - sg);
      }
      *reinterpret_cast<uint4*>(at(dst, c0)) = cl_pack8(ga);
      *reinterpret_cast<uint4*>(at(dst, C + c0)) = cl_pack8(gb);
    }
  }
}

// aux16 (optional, bf16 channel-major with dst's strides): dst = v * gelu'(aux16) -- the backward of a GELU whose input is stored
// channel-major (the first encoder layer's convolution output)
template <typename T>
__global__ __launch_bounds__(256) void cl_to_cm_kernel(const uint16_t* __restrict__ src, int64_t s_ns, int64_t s_as, int s_bs, int C, int A,
                                                       int B, T* __restrict__ dst, int64_t d_ns, int64_t d_cs, int64_t d_as,
                                                       const rfx_bf16s* __restrict__ aux16) {
  __shared__ uint16_t tile[32][66];
  const int tpb = B / 64;
  const int pt = blockIdx.x, cgp = blockIdx.y;
  const int n = pt / (A * tpb), rem = pt - n * A * tpb, a = rem / tpb, b0 = (rem - a * tpb) * 64;
  const int t = threadIdx.x;
  const int pp = t >> 2, g8 = t & 3, c0 = cgp * 32 + g8 * 8;
  if (c0 < C) {
    const uint4 u = *reinterpret_cast<const uint4*>(src + (int64_t)n * s_ns + (int64_t)a * s_as + (int64_t)(b0 + pp) * s_bs + c0);
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      tile[g8 * 8 + 2 * e][pp] = (uint16_t)(w[e] & 0xffffu);
      tile[g8 * 8 + 2 * e + 1][pp] = (uint16_t)(w[e] >> 16);
    }
  }
  __syncthreads();
  const int p = t & 63, cr = t >> 6;
  const int64_t ob = (int64_t)n * d_ns + (int64_t)a * d_as + b0 + p;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = cr + 4 * k, cc = cgp * 32 + c;
    if (cc < C) {
      float v = cl_bf2f(tile[c][p]);
      if (aux16 != nullptr) v *= rfx_gelu_grad(rfx_ld1(aux16 + ob + (int64_t)cc * d_cs));
      rfx_st1(dst + ob + (int64_t)cc * d_cs, v);
    }
  }
}

template <typename T>
static void cl_from_cm_launch(int mode, dim3 grid, hipStream_t st, const T* src, int64_t s_ns, int64_t s_cs, int64_t s_as, int C, int A,
                              int B, const rfx_cl_tensor& dst, const rfx_cl_tensor& res, const rfx_cl_tensor& aux) {
  if (mode == 0) hipLaunchKernelGGL((cl_from_cm_kernel<T, 0>), grid, dim3(256), 0, st, src, s_ns, s_cs, s_as, C, A, B, dst, res, aux);
  else if (mode == 1) hipLaunchKernelGGL((cl_from_cm_kernel<T, 1>), grid, dim3(256), 0, st, src, s_ns, s_cs, s_as, C, A, B, dst, res, aux);
  else if (mode == 3) hipLaunchKernelGGL((cl_from_cm_kernel<T, 3>), grid, dim3(256), 0, st, src, s_ns, s_cs, s_as, C, A, B, dst, res, aux);
  else hipLaunchKernelGGL((cl_from_cm_kernel<T, 2>), grid, dim3(256), 0, st, src, s_ns, s_cs, s_as, C, A, B, dst, res, aux);
}

extern "C" int rfx_cl_from_cm(const void* src, int32_t src_bf16, int64_t s_ns, int64_t s_cs, int64_t s_as, int32_t N, int32_t C,
                              int32_t A, int32_t B, const rfx_cl_tensor* dst, const rfx_cl_tensor* res, const rfx_cl_tensor* aux,
                              int32_t mode, void* stream) {
  if (!src || !dst || !dst->p || N <= 0 || C <= 0 || A <= 0 || B <= 0 || B % 64 || C % 8 || dst->c0 % 8 || dst->bs % 8) return -1;
  if (mode < 0 || mode > 3 || ((mode == 1 || mode == 2) && (!aux || !aux->p || aux->bs % 8 || aux->c0 % 8))) return -1;
  if (res && res->p && (res->bs % 8 || res->c0 % 8)) return -1;
  if (mode == 2 && (dst->bs < 2 * C || aux->bs < 2 * C)) return -1;
  const dim3 grid((unsigned)(N * A * (B / 64)), (unsigned)((C + 31) / 32));
  const rfx_cl_tensor none = {nullptr, 0, 0, 0, 0};
  const rfx_cl_tensor& r = (res && res->p) ? *res : none;
  const rfx_cl_tensor& x = (aux && aux->p) ? *aux : none;
  if (src_bf16)
    cl_from_cm_launch(mode, grid, (hipStream_t)stream, reinterpret_cast<const rfx_bf16s*>(src), s_ns, s_cs, s_as, C, A, B, *dst, r, x);
  else
    cl_from_cm_launch(mode, grid, (hipStream_t)stream, reinterpret_cast<const float*>(src), s_ns, s_cs, s_as, C, A, B, *dst, r, x);
  RFX_CHECK_LAUNCH();
  return 0;
}

extern "C" int rfx_cl_to_cm(const rfx_cl_tensor* src, int32_t N, int32_t C, int32_t A, int32_t B, void* dst, int32_t dst_bf16,
                            int64_t d_ns, int64_t d_cs, int64_t d_as, const void* aux16, void* stream) {
  if (!src || !src->p || !dst || N <= 0 || C <= 0 || A <= 0 || B <= 0 || B % 64 || C % 8 || src->c0 % 8 || src->bs % 8) return -1;
  const dim3 grid((unsigned)(N * A * (B / 64)), (unsigned)((C + 31) / 32));
  const uint16_t* s = reinterpret_cast<const uint16_t*>(src->p) + src->c0;
  if (dst_bf16)
    hipLaunchKernelGGL(cl_to_cm_kernel<rfx_bf16s>, grid, dim3(256), 0, (hipStream_t)stream, s, src->ns, src->as, src->bs, C, A, B,
                       reinterpret_cast<rfx_bf16s*>(dst), d_ns, d_cs, d_as, reinterpret_cast<const rfx_bf16s*>(aux16));
  else
    hipLaunchKernelGGL(cl_to_cm_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, s, src->ns, src->as, src->bs, C, A, B,
                       reinterpret_cast<float*>(dst), d_ns, d_cs, d_as, reinterpret_cast<const rfx_bf16s*>(aux16));
  RFX_CHECK_LAUNCH();
  return 0;
}


// ---- deterministic sums over positions ------------------------------------------------------------------------------------------
// out[a][c] (+)= scale * sum over (n, b) of x[n][a][b][c]  (A = 1 with the rows folded into N: the per-channel sum = a bias
// gradient; A = frequency rows: the gradient of the frequency embedding, torchaudio HDemucs `freq_emb`).  Stage 1: block (a, g)
// sums the samples n = g, g + G, ... in a fixed order into partial[g][a][c]; stage 2 adds the G partials in order.  No atomics.
__global__ __launch_bounds__(256) void cl_rowsum_partial_kernel(rfx_cl_tensor x, int N, int A, int B, int C, int G, float* __restrict__ partial) {
  __shared__ float red[256][8];
  const int a = blockIdx.x, g = blockIdx.y, t = threadIdx.x;
  const int CG = C >> 3, PT = 256 / CG;                      // channel groups of 8, positions walked in parallel
  const int grp = t % CG, pl = t / CG;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (pl < PT) {
    for (int n = g; n < N; n += G) {
      const uint16_t* row = reinterpret_cast<const uint16_t*>(x.p) + (int64_t)n * x.ns + (int64_t)a * x.as + x.c0 + grp * 8;
      for (int b = pl; b < B; b += PT) {
        float v[8];
        cl_unpack8(*reinterpret_cast<const uint4*>(row + (int64_t)b * x.bs), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += v[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[t][e] = acc[e];
  __syncthreads();
  if (t < C) {                                               // channel t: groups' partials of the PT position lanes, in order
    const int gq = t >> 3, e = t & 7;
    float s = 0.f;
    for (int q = 0; q < PT; ++q) s += red[q * CG + gq][e];
    partial[((int64_t)g * A + a) * C + t] = s;
  }
}
// one wave per output: lane l adds partials l, l + 64, ... in order, then a fixed butterfly over the lanes
__global__ __launch_bounds__(256) void cl_rowsum_final_kernel(const float* __restrict__ partial, int64_t n, int G, float scale,
                                                              float* __restrict__ out, int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int l = threadIdx.x & 63;
  if (i >= n) return;
  float s = 0.f;
  for (int g = l; g < G; g += 64) s += partial[(int64_t)g * n + i];
  s = rfx_wave_sum(s) * scale;
  if (l == 0) out[i] = accumulate ? out[i] + s : s;
}

extern "C" int rfx_cl_rowsum(const rfx_cl_tensor* x, int32_t N, int32_t A, int32_t B, int32_t C, int32_t G, float scale, float* partial,
                             float* out, int32_t accumulate, void* stream) {
  if (!x || !x->p || !partial || !out || N <= 0 || A <= 0 || B <= 0 || C <= 0 || C % 8 || C > 256 || G < 1 || x->bs % 8 || x->c0 % 8)
    return -1;
  hipLaunchKernelGGL(cl_rowsum_partial_kernel, dim3((unsigned)A, (unsigned)G), dim3(256), 0, (hipStream_t)stream, *x, N, A, B, C, G, partial);
  const int64_t n = (int64_t)A * C;
  hipLaunchKernelGGL(cl_rowsum_final_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, partial, n, G, scale, out,
                     accumulate);
  RFX_CHECK_LAUNCH();
  return 0;
}


// ---- flat elementwise on dense channels-last tensors -----------------------------------------------------------------------------
// out = g * gelu'(z) over n8 groups of 8 bf16 (the backward of a GELU between two channels-last nodes)
__global__ __launch_bounds__(256) void cl_dgelu_kernel(const uint4* __restrict__ g, const uint4* __restrict__ z, uint4* __restrict__ out, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    float a[8], b[8], o[8];
    cl_unpack8(g[i], a);
    cl_unpack8(z[i], b);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = a[e] * rfx_gelu_grad(b[e]);
    out[i] = cl_pack8(o);
  }
}
extern "C" int rfx_cl_dgelu(const void* g, const void* z, void* out, int64_t n, void* stream) {
  if (!g || !z || !out || n <= 0 || n % 8) return -1;
  const int64_t n8 = n / 8;
  const int grid = (int)((n8 + 255) / 256 < 16384 ? (n8 + 255) / 256 : 16384);
  hipLaunchKernelGGL(cl_dgelu_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const uint4*>(g),
                     reinterpret_cast<const uint4*>(z), reinterpret_cast<uint4*>(out), n8);
  RFX_CHECK_LAUNCH();
  return 0;
}

// out [n][2 C] = GLU backward of g [n][C] against the stored zab [n][2 C] = [a | b]: [g * sigmoid(b) | g * a * sigmoid(b) (1 - sigmoid(b))]
__global__ __launch_bounds__(256) void cl_dglu_kernel(const uint4* __restrict__ g, const uint4* __restrict__ zab, uint4* __restrict__ out, int64_t npos, int CG) {
  const int64_t total = npos * CG;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t p = i / CG;
    const int cg = (int)(i - p * CG);
    float v[8], a[8], b[8], ga[8], gb[8];
    cl_unpack8(g[i], v);
    cl_unpack8(zab[p * 2 * CG + cg], a);
    cl_unpack8(zab[p * 2 * CG + CG + cg], b);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float sg = rfx_sigmoid(b[e]);
      ga[e] = v[e] * sg;
      gb[e] = v[e] * a[e] * sg * (1.f - sg);
    }
    out[p * 2 * CG + cg] = cl_pack8(ga);
    out[p * 2 * CG + CG + cg] = cl_pack8(gb);
  }
}
extern "C" int rfx_cl_dglu(const void* g, const void* zab, void* out, int64_t npos, int32_t C, void* stream) {
  if (!g || !zab || !out || npos <= 0 || C <= 0 || C % 8) return -1;
  const int64_t total = npos * (C / 8);
  const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipLaunchKernelGGL(cl_dglu_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const uint4*>(g),
                     reinterpret_cast<const uint4*>(zab), reinterpret_cast<uint4*>(out), npos, C / 8);
  RFX_CHECK_LAUNCH();
  return 0;
}

// ---- the 16-channel operand of the network's first convolution / last transposed convolution (see remfx_hip.h)
__global__ __launch_bounds__(256) void cl_im2col_s4_kernel(const float* __restrict__ src, int64_t s_ns, int64_t s_cs, int64_t s_as, int N, int Cs,
                                                           int IA, int IB, int OA, int OB, int along_b, uint4* __restrict__ dst) {
  const int64_t total = (int64_t)N * OA * OB;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int b = (int)(i % OB);
    const int64_t q = i / OB;
    const int oa = (int)(q % OA), n = (int)(q / OA);
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = 0.f;
    const float* s = src + (int64_t)n * s_ns;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int ia = along_b ? oa : 4 * oa + k - 2, ib = along_b ? 4 * b + k - 2 : b;
      if ((unsigned)ia < (unsigned)IA && (unsigned)ib < (unsigned)IB) {
        v[k * Cs] = s[(int64_t)ia * s_as + ib];
        if (Cs == 2) v[k * 2 + 1] = s[s_cs + (int64_t)ia * s_as + ib];
      }
    }
    float lo[8], hi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { lo[e] = v[e]; hi[e] = v[8 + e]; }
    dst[2 * i] = cl_pack8(lo);
    dst[2 * i + 1] = cl_pack8(hi);
  }
}
extern "C" int rfx_cl_im2col_s4(const float* src, int64_t s_ns, int64_t s_cs, int64_t s_as, int32_t N, int32_t Cs, int32_t IA, int32_t IB, int32_t OA,
                                int32_t OB, int32_t along_b, void* dst, void* stream) {
  if (!src || !dst || N <= 0 || (Cs != 1 && Cs != 2) || IA <= 0 || IB <= 0 || OA <= 0 || OB <= 0) return -1;
  const int64_t total = (int64_t)N * OA * OB;
  const int grid = (int)((total + 255) / 256 < 32768 ? (total + 255) / 256 : 32768);
  hipLaunchKernelGGL(cl_im2col_s4_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, s_ns, s_cs, s_as, N, Cs, IA, IB, OA, OB, along_b,
                     reinterpret_cast<uint4*>(dst));
  RFX_CHECK_LAUNCH();
  return 0;
}

// ---- frame-major ends of the Hybrid Demucs frequency branch (round 6) ---------------------------------------------------------------
// The STFT kernels store and load a frame's bins as one contiguous run when the spectrum is frame-major ([R][frames][bins][2],
// RFX_STFT_COMPLEX_FM); the [bin][frame] layout of torch.stft costs them 8-byte pieces of 128-byte lines.  The two
// kernels below are what lets HDemucs keep its spectrum frame-major on both sides of the U-Net.
//
// (1) The first convolution's 16-channel im2col operand straight from the frame-major spectrum, with the per-clip standardisation
// (x a[n] + b[n]) applied on the way: element k * 2 + c of output position (n, oa, f) = a x[n][f][4 oa + k - 2][c] + b, zero where the bin
// is outside [0, bins) (the convolution's zero padding applies to the STANDARDISED tensor).  The 16 values are 64 contiguous bytes of
// the source.  A workgroup transposes a tile of 32 frames x 32 output rows through LDS: reads run along bins, writes along frames.
__global__ __launch_bounds__(256) void cl_im2col_fm_kernel(const float* __restrict__ src, const float* __restrict__ ca, const float* __restrict__ cb,
                                                           int F, int bins, int OA, uint4* __restrict__ dst) {
  constexpr int TF = 32, TA = 32, ROWF = 4 * TA * 2 + 8;         // floats of one frame's slab: bins [4 oa0 - 2, 4 oa0 + 4 TA + 2) x (re, im)
  __shared__ __attribute__((aligned(16))) float slab[TF][ROWF + 4];
  const int n = blockIdx.z, f0 = blockIdx.y * TF, oa0 = blockIdx.x * TA;
  const float a = ca[n], b = cb[n];
  const float* s = src + (int64_t)n * F * bins * 2;
  const int bin0 = 4 * oa0 - 2;
  for (int i = threadIdx.x; i < TF * (ROWF / 4); i += 256) {     // float4 = two bins
    const int fl = i / (ROWF / 4), q = i - fl * (ROWF / 4);
    const int bin = bin0 + 2 * q, f = f0 + fl;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (f < F && bin >= 0 && bin + 1 < bins + 1 && bin < bins) {
      v = *reinterpret_cast<const f32x4*>(s + ((int64_t)f * bins + bin) * 2);
      v[0] = fmaf(v[0], a, b); v[1] = fmaf(v[1], a, b); v[2] = fmaf(v[2], a, b); v[3] = fmaf(v[3], a, b);
    }
    *reinterpret_cast<f32x4*>(&slab[fl][4 * q]) = v;
  }
  __syncthreads();
  const int fl = threadIdx.x & 31;
  for (int al = threadIdx.x >> 5; al < TA; al += 8) {
    const int oa = oa0 + al, f = f0 + fl;
    if (oa >= OA || f >= F) continue;
    float lo[8], hi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { lo[e] = slab[fl][8 * al + e]; hi[e] = slab[fl][8 * al + 8 + e]; }
    const int64_t i = ((int64_t)n * OA + oa) * F + f;
    dst[2 * i] = cl_pack8(lo);
    dst[2 * i + 1] = cl_pack8(hi);
  }
}
extern "C" int rfx_cl_im2col_fm(const float* src, const float* coef_a, const float* coef_b, int32_t N, int32_t F, int32_t bins, void* dst,
                                void* stream) {
  if (!src || !coef_a || !coef_b || !dst || N <= 0 || F <= 0 || bins <= 0 || (bins & 3) || (reinterpret_cast<uintptr_t>(src) & 15)) return -1;
  const int OA = bins / 4;
  hipLaunchKernelGGL(cl_im2col_fm_kernel, dim3((OA + 31) / 32, (F + 31) / 32, N), dim3(256), 0, (hipStream_t)stream, src, coef_a, coef_b, F,
                     bins, OA, reinterpret_cast<uint4*>(dst));
  RFX_CHECK_LAUNCH();
  return 0;
}

// (2) Layout change between the channel-major spectrum of the last transposed convolution, x[n][c][bin][frame], and the frame-major
// one the inverse STFT reads, y[n][frame][bin][c], fused with the per-clip affine map of HDemucs' de-standardisation (y = x a[n] + b[n];
// b == NULL: scale only -- the backward direction, to_fm = 0: x[n][c][bin][frame] = y[n][frame][bin][c] a[n]).  32 x 32 x 2 tiles through
// LDS: both sides move full 128 / 256-byte runs.
__global__ __launch_bounds__(256) void fm_cm_affine_kernel(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ ca,
                                                           const float* __restrict__ cb, int bins, int F, int to_fm) {
  __shared__ float tile[2][32][33];
  const int n = blockIdx.z, k0 = blockIdx.x * 32, f0 = blockIdx.y * 32;
  const float a = ca[n], b = cb ? cb[n] : 0.f;
  const int64_t cm_n = (int64_t)n * 2 * bins * F, fm_n = (int64_t)n * F * bins * 2;
  if (to_fm) {
    for (int i = threadIdx.x; i < 2048; i += 256) {
      const int c = i >> 10, k = (i >> 5) & 31, f = i & 31;
      tile[c][k][f] = (k0 + k < bins && f0 + f < F) ? fmaf(in[cm_n + ((int64_t)c * bins + k0 + k) * F + f0 + f], a, b) : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 256) {
      const int f = i >> 6, k = (i >> 1) & 31, c = i & 1;
      if (k0 + k < bins && f0 + f < F) out[fm_n + ((int64_t)(f0 + f) * bins + k0 + k) * 2 + c] = tile[c][k][f];
    }
  } else {
    for (int i = threadIdx.x; i < 2048; i += 256) {
      const int f = i >> 6, k = (i >> 1) & 31, c = i & 1;
      tile[c][k][f] = (k0 + k < bins && f0 + f < F) ? fmaf(in[fm_n + ((int64_t)(f0 + f) * bins + k0 + k) * 2 + c], a, b) : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 256) {
      const int c = i >> 10, k = (i >> 5) & 31, f = i & 31;
      if (k0 + k < bins && f0 + f < F) out[cm_n + ((int64_t)c * bins + k0 + k) * F + f0 + f] = tile[c][k][f];
    }
  }
}
extern "C" int rfx_fm_cm_affine(const float* in, float* out, const float* coef_a, const float* coef_b, int32_t N, int32_t bins, int32_t F,
                                int32_t to_fm, void* stream) {
  if (!in || !out || !coef_a || N <= 0 || bins <= 0 || F <= 0) return -1;
  hipLaunchKernelGGL(fm_cm_affine_kernel, dim3((bins + 31) / 32, (F + 31) / 32, N), dim3(256), 0, (hipStream_t)stream, in, out, coef_a, coef_b,
                     bins, F, to_fm);
  RFX_CHECK_LAUNCH();
  return 0;
}
