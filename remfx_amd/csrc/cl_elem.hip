// Channels-last bf16 family: layout conversion at the ends of the channels-last trunk (the spectrogram branch of Hybrid Demucs
// enters from the STFT / the merged deep layers in channel-major fp32 and leaves the same way; torchaudio HDemucs behind
// remfx/models.py:308,317) and the small elementwise kernels of that trunk.
#include "cl_common.h"

// (N, C, A, B) channel-major, B contiguous  ->  [N][A][B][bs] channels-last bf16.  Tile = 64 positions x 32 channels through LDS.
template <typename T>
__global__ __launch_bounds__(256) void cl_from_cm_kernel(const T* __restrict__ src, int64_t s_ns, int64_t s_cs, int64_t s_as, int C,
                                                         int A, int B, uint16_t* __restrict__ dst, int64_t d_ns, int64_t d_as, int d_bs) {
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
    *reinterpret_cast<uint4*>(dst + (int64_t)n * d_ns + (int64_t)a * d_as + (int64_t)(b0 + pp) * d_bs + c0) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void cl_to_cm_kernel(const uint16_t* __restrict__ src, int64_t s_ns, int64_t s_as, int s_bs, int C, int A,
                                                       int B, T* __restrict__ dst, int64_t d_ns, int64_t d_cs, int64_t d_as) {
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
  T* o = dst + (int64_t)n * d_ns + (int64_t)a * d_as + b0 + p;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = cr + 4 * k, cc = cgp * 32 + c;
    if (cc < C) rfx_st1(o + (int64_t)cc * d_cs, cl_bf2f(tile[c][p]));
  }
}

extern "C" int rfx_cl_from_cm(const void* src, int32_t src_bf16, int64_t s_ns, int64_t s_cs, int64_t s_as, int32_t N, int32_t C,
                              int32_t A, int32_t B, const rfx_cl_tensor* dst, void* stream) {
  if (!src || !dst || !dst->p || N <= 0 || C <= 0 || A <= 0 || B <= 0 || B % 64 || C % 8 || dst->c0 % 8 || dst->bs % 8) return -1;
  const dim3 grid((unsigned)(N * A * (B / 64)), (unsigned)((C + 31) / 32));
  uint16_t* d = reinterpret_cast<uint16_t*>(dst->p) + dst->c0;
  if (src_bf16)
    hipLaunchKernelGGL(cl_from_cm_kernel<rfx_bf16s>, grid, dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const rfx_bf16s*>(src),
                       s_ns, s_cs, s_as, C, A, B, d, dst->ns, dst->as, dst->bs);
  else
    hipLaunchKernelGGL(cl_from_cm_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float*>(src), s_ns,
                       s_cs, s_as, C, A, B, d, dst->ns, dst->as, dst->bs);
  RFX_CHECK_LAUNCH();
  return 0;
}

extern "C" int rfx_cl_to_cm(const rfx_cl_tensor* src, int32_t N, int32_t C, int32_t A, int32_t B, void* dst, int32_t dst_bf16,
                            int64_t d_ns, int64_t d_cs, int64_t d_as, void* stream) {
  if (!src || !src->p || !dst || N <= 0 || C <= 0 || A <= 0 || B <= 0 || B % 64 || C % 8 || src->c0 % 8 || src->bs % 8) return -1;
  const dim3 grid((unsigned)(N * A * (B / 64)), (unsigned)((C + 31) / 32));
  const uint16_t* s = reinterpret_cast<const uint16_t*>(src->p) + src->c0;
  if (dst_bf16)
    hipLaunchKernelGGL(cl_to_cm_kernel<rfx_bf16s>, grid, dim3(256), 0, (hipStream_t)stream, s, src->ns, src->as, src->bs, C, A, B,
                       reinterpret_cast<rfx_bf16s*>(dst), d_ns, d_cs, d_as);
  else
    hipLaunchKernelGGL(cl_to_cm_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, s, src->ns, src->as, src->bs, C, A, B,
                       reinterpret_cast<float*>(dst), d_ns, d_cs, d_as);
  RFX_CHECK_LAUNCH();
  return 0;
}
