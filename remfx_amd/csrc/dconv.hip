// Fused DConv depth-layer of the Hybrid Demucs frequency branch (torchaudio HDemucs `_DConv`, reached from
// remfx/models.py:319), bf16 arithmetic mode:
//     h = conv1d(x; W1 (H, C, 3), b1, dilation d, padding d)            H = C / 4
//     a = GELU(GroupNorm(1, H)(h))
//     z = conv1d(a; W2 (2C, H, 1), b2)
//     x_out = x + scale[c] * GLU(GroupNorm(1, 2C)(z))
// for N independent samples of (C, T = 256) -- the frequency branch runs its DConv over (batch x frequency) rows of 256
// frames, so one sample (48 KB at C = 48) fits a workgroup: ONE pass over x per direction instead of the eleven the
// layer-by-layer path makes (conv, norm, conv, norm + GLU + residual; their backward kernels and the two weight-gradient
// GEMMs).  One workgroup of 4 waves walks over samples (persistent grid); wave w owns positions [64 w, 64 w + 64) = two
// 32-column MFMA tiles and all rows.  Both GEMMs run on v_mfma_f32_32x32x16_bf16:
//   * GEMM1's B operand (x at the three taps) comes from a channels-last bf16 image of the sample in LDS: one
//     ds_read_b128 per K step; its A operand (W1) is packed once per workgroup into MFMA fragments in LDS;
//   * the 32x32 C/D layout (lane (j, h): rows (r&3) + 8(r>>2) + 4h of column j) is a valid B operand of the next MFMA when
//     the A fragments enumerate k in that order, so GELU(GN(h)) feeds GEMM2 straight from registers;
//   * W2's rows are permuted so that tile mt holds the GLU "value" rows of channels 16mt..16mt+15 in its first 16 rows and
//     their "gate" rows in the last 16: both halves of a GLU pair then sit in ONE lane (registers r and r + 8), and the
//     residual x is loaded in the same register layout.
// GroupNorm(1, .) statistics are sample-wide: per-wave partial sums meet in LDS (three workgroup barriers per sample).
// In training the kernel also stores what the layer-by-layer BACKWARD kernels read (h and z as bf16 -- what the unfused path
// stores in this mode -- the GELU output a, and both GroupNorm (mean, rstd) pairs), so the backward pass runs on the existing
// GroupNorm / input-gradient / weight-gradient kernels.  (A fused backward kernel -- forward recomputed, parameter gradients in
// per-lane registers -- was built and verified, but its 250 live values per lane spill at one wave per SIMD: 9.4 ms per layer
// against 4.8 ms layer-by-layer at N = 32768, so it is not in the library)
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define DC_T 256
#define DC_PADR 2      // zero rows either side of the channels-last image (largest dilation of the two depth layers)

struct DconvArgs {
  const float* x;        // (N, C, T)
  float* out;            // (N, C, T)
  const float *w1, *b1, *gn1w, *gn1b, *w2, *b2, *gn2w, *gn2b, *scale;
  // training: what the backward kernels read (all NULL in inference)
  uint16_t* h16;         // (N, H, T) bf16: conv1 output (+ bias)
  uint16_t* z16;         // (N, 2C, T) bf16: conv2 output (+ bias), natural row order
  float* a_out;          // (N, H, T): GELU(GroupNorm(h))
  float* stats;          // (4, N): mean1, rstd1, mean2, rstd2
  int N, dil;
  float eps;
};

template <int C>
struct DcCfg {
  static constexpr int H = C / 4, G8 = C / 8, NK1 = 3 * G8 / 2, MT = C / 16, NK2 = (H + 15) / 16, CP = C + 8;
};

// GELU / GELU' / sigmoid: the branch-free forms of common.h (Abramowitz-Stegun erf, v_exp + v_rcp)
__device__ __forceinline__ void dc_gelu_parts(float x, float& cdf, float& ex) { rfx_gelu_parts(x, cdf, ex); }
__device__ __forceinline__ float dc_gelu(float x) { return rfx_gelu(x); }
__device__ __forceinline__ float dc_gelu_grad(float x) { return rfx_gelu_grad(x); }
__device__ __forceinline__ float dc_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// one sample of a (rows, T) tensor as a buffer: per-lane byte offset (row of lane half, column) + a compile-time row offset in
// the scalar offset field -> no per-access address arithmetic
template <typename T>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dc_rsrc(const T* base, int64_t sample, int rows) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base + sample * rows * DC_T), 0, rows * DC_T * (int)sizeof(T), 0x00020000);
}
// byte offset of row (16 mt + (r&3) + 8 (r>>2)) of a tensor with ESZ-byte elements (the 4 hh rows and the column are in the lane offset)
#define DC_ROFF(mt, r, esz) ((16 * (mt) + ((r) & 3) + 8 * ((r) >> 2)) * DC_T * (esz))

__device__ __forceinline__ uint32_t dc_pack2(float a, float b) { return rfx_bf16_bits(a) | (rfx_bf16_bits(b) << 16); }
__device__ __forceinline__ bf16x8 dc_frag8(const float* v) {     // 8 floats -> bf16x8 (RNE)
  return __builtin_bit_cast(bf16x8, make_uint4(dc_pack2(v[0], v[1]), dc_pack2(v[2], v[3]), dc_pack2(v[4], v[5]), dc_pack2(v[6], v[7])));
}
// channel of register r (< 8) of 16-row half-tile mt for lane half hh
__device__ __forceinline__ int dc_chan(int mt, int r, int hh) { return 16 * mt + (r & 3) + 8 * (r >> 2) + 4 * hh; }

// shared-memory image of one workgroup
template <int C>
struct DcLds {
  using K = DcCfg<C>;
  uint16_t xs[(DC_T + 2 * DC_PADR) * K::CP];       // channels-last bf16 image of the sample, zero rows either side
  uint4 w1f[K::NK1 * 64];                          // A fragments of W1, K order (tap, 8-channel group)
  uint4 w2f[K::MT * K::NK2 * 64];                  // A fragments of the row-permuted W2, K order = C/D register order
  float b1[32], g1[32], be1[32];                   // conv1 bias, GroupNorm(1, H) affine, by hidden row (0 beyond H)
  float pz[7][K::MT * 16];                         // b2 / gn2w / gn2b of the value and gate rows + scale, in (mt, hh, r) lane order
  float red[4][4];
};

template <int C>
__device__ __forceinline__ void dc_load_params(DcLds<C>& s, const DconvArgs& a, int tid) {
  using K = DcCfg<C>;
  for (int idx = tid; idx < K::NK1 * 64; idx += 256) {
    const int ks = idx >> 6, lane = idx & 63, i = lane & 31, hh = lane >> 5;
    const int g = 2 * ks + hh, tap = g / K::G8, cg = g % K::G8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = i < K::H ? a.w1[((int64_t)i * C + cg * 8 + e) * 3 + tap] : 0.f;
    s.w1f[idx] = __builtin_bit_cast(uint4, dc_frag8(v));
  }
  for (int idx = tid; idx < K::MT * K::NK2 * 64; idx += 256) {
    const int lane = idx & 63, ks = (idx >> 6) % K::NK2, mt = (idx >> 6) / K::NK2, i = lane & 31, hh = lane >> 5;
    const int row = i < 16 ? 16 * mt + i : C + 16 * mt + (i - 16);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 16 * ks + (e & 3) + 8 * (e >> 2) + 4 * hh;
      v[e] = k < K::H ? a.w2[(int64_t)row * K::H + k] : 0.f;
    }
    s.w2f[idx] = __builtin_bit_cast(uint4, dc_frag8(v));
  }
  if (tid < 32) {
    s.b1[tid] = tid < K::H ? a.b1[tid] : 0.f;
    s.g1[tid] = tid < K::H ? a.gn1w[tid] : 0.f;
    s.be1[tid] = tid < K::H ? a.gn1b[tid] : 0.f;
  }
  for (int idx = tid; idx < K::MT * 16; idx += 256) {
    const int mt = idx >> 4, hh = (idx >> 3) & 1, r = idx & 7, c = dc_chan(mt, r, hh);
    s.pz[0][idx] = a.b2[c];     s.pz[1][idx] = a.b2[C + c];
    s.pz[2][idx] = a.gn2w[c];   s.pz[3][idx] = a.gn2w[C + c];
    s.pz[4][idx] = a.gn2b[c];   s.pz[5][idx] = a.gn2b[C + c];
    s.pz[6][idx] = a.scale[c];
  }
  for (int idx = tid; idx < 2 * DC_PADR * K::CP; idx += 256) {      // zero rows: written once
    const int row = idx / K::CP, col = idx % K::CP;
    s.xs[(row < DC_PADR ? row : DC_T + row) * K::CP + col] = 0;
  }
  for (int idx = tid; idx < DC_T * 8; idx += 256)                  // channel padding columns (never read by GEMM1; keep defined)
    s.xs[(DC_PADR + idx / 8) * K::CP + C + idx % 8] = 0;
}

// x registers of one 32-column tile in the (mt, r) layout + the channels-last bf16 image.  voff = (4 hh * T + t) * 4
template <int C>
__device__ __forceinline__ void dc_load_x(__amdgpu_buffer_rsrc_t rs, uint32_t voff, float (&xr)[DcCfg<C>::MT][8]) {
#pragma unroll
  for (int mt = 0; mt < DcCfg<C>::MT; ++mt)
#pragma unroll
    for (int r = 0; r < 8; ++r) xr[mt][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, DC_ROFF(mt, r, 4), 0));
}
template <int C>
__device__ __forceinline__ void dc_store_xs(DcLds<C>& s, int t, int hh, const float (&xr)[DcCfg<C>::MT][8]) {
  using K = DcCfg<C>;
  uint16_t* row = s.xs + (t + DC_PADR) * K::CP;
#pragma unroll
  for (int mt = 0; mt < K::MT; ++mt)
#pragma unroll
    for (int q = 0; q < 2; ++q)
      *reinterpret_cast<uint2*>(row + 16 * mt + 8 * q + 4 * hh) =
          make_uint2(dc_pack2(xr[mt][4 * q], xr[mt][4 * q + 1]), dc_pack2(xr[mt][4 * q + 2], xr[mt][4 * q + 3]));
}

// GEMM1 of one 32-column tile: h[m][t] = sum_{tap, c} W1[m][c][tap] * x[c][t + (tap - 1) dil]   (bias added by the caller)
template <int C>
__device__ __forceinline__ f32x16 dc_gemm1(const DcLds<C>& s, int t, int hh, int lane, int dil) {
  using K = DcCfg<C>;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < K::NK1; ++ks) {
    const int tap = (2 * ks) / K::G8, cg = (2 * ks) % K::G8;          // compile-time per ks; lane half hh takes group cg + 1 (G8 is even)
    const uint16_t* p = s.xs + (t + DC_PADR + (tap - 1) * dil) * K::CP + (cg + hh) * 8;
    const bf16x8 b = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(p));
    const bf16x8 af = __builtin_bit_cast(bf16x8, s.w1f[ks * 64 + lane]);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b, acc, 0, 0, 0);
  }
  return acc;
}
// hidden row of accumulator register r of lane half hh
__device__ __forceinline__ int dc_hrow(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

__device__ __forceinline__ void dc_wg_sum2(float (&red)[4][4], int wave, int lane, float& a, float& b, int slot) {
  a = rfx_wave_sum(a); b = rfx_wave_sum(b);
  if (lane == 0) { red[wave][slot] = a; red[wave][slot + 1] = b; }
  __syncthreads();
  a = red[0][slot] + red[1][slot] + red[2][slot] + red[3][slot];
  b = red[0][slot + 1] + red[1][slot + 1] + red[2][slot + 1] + red[3][slot + 1];
}

template <int C, bool SAVE>
__global__ __launch_bounds__(256, 2) void dconv_fwd_kernel(const DconvArgs a) {
  using K = DcCfg<C>;
  extern __shared__ __attribute__((aligned(16))) unsigned char dc_smem[];
  DcLds<C>& s = *reinterpret_cast<DcLds<C>*>(dc_smem);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, hh = lane >> 5;
  dc_load_params<C>(s, a, tid);
  const uint32_t voff0 = (uint32_t)((4 * hh * DC_T + 64 * wave + j) * 4);           // column tile ct adds 32 * 4 bytes
  const float inv1 = 1.0f / (float)(K::H * DC_T), inv2 = 1.0f / (float)(2 * C * DC_T);
  for (int n = blockIdx.x; n < a.N; n += gridDim.x) {
    const __amdgpu_buffer_rsrc_t xrs = dc_rsrc(a.x, n, C), ors = dc_rsrc(a.out, n, C);
    __amdgpu_buffer_rsrc_t hrs = xrs, zrs = xrs, ars = xrs;
    if (SAVE) { hrs = dc_rsrc(a.h16, n, K::H); zrs = dc_rsrc(a.z16, n, 2 * C); ars = dc_rsrc(a.a_out, n, K::H); }
    float xr[2][K::MT][8];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) dc_load_x<C>(xrs, voff0 + 128 * ct, xr[ct]);      // two workgroups per CU cover each other's latency
    __syncthreads();                       // everybody is done with the previous sample's image (and the parameter tables are in)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) dc_store_xs<C>(s, 64 * wave + 32 * ct + j, hh, xr[ct]);
    __syncthreads();
    // ---- GEMM1 + GroupNorm(1, H) statistics
    f32x16 hacc[2];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      hacc[ct] = dc_gemm1<C>(s, 64 * wave + 32 * ct + j, hh, lane, a.dil);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = dc_hrow(r, hh);
        const float v = hacc[ct][r] + s.b1[m];
        hacc[ct][r] = v;
        s1 += v; s2 = fmaf(v, v, s2);                      // rows beyond H are exactly 0 (zero weight rows, zero bias): no mask
        // rows beyond H fall outside the (H, T) buffer: the hardware range check drops those stores
        if (SAVE) __builtin_amdgcn_raw_buffer_store_b16((short)rfx_bf16_bits(v), hrs, (voff0 >> 1) + 64 * ct, ((r & 3) + 8 * (r >> 2)) * DC_T * 2, 0);
      }
    }
    dc_wg_sum2(s.red, wave, lane, s1, s2, 0);
    const float mu1 = s1 * inv1, rs1 = rsqrtf(fmaxf(s2 * inv1 - mu1 * mu1, 0.f) + a.eps);
    // ---- a = GELU(gn(h)) -> B fragments of GEMM2 straight from registers; GEMM2; GroupNorm(1, 2C) statistics
    f32x16 z[2][K::MT];
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      bf16x8 af[K::NK2];
#pragma unroll
      for (int ks = 0; ks < K::NK2; ++ks) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int m = 16 * ks + dc_hrow(e, hh);
          const float hn = (hacc[ct][8 * ks + e] - mu1) * rs1 * s.g1[m] + s.be1[m];
          v[e] = dc_gelu(hn);                              // rows beyond H: zero affine -> hn = 0 -> GELU(0) = 0, branch-free
          if (SAVE) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v[e]), ars, voff0 + 128 * ct, (16 * ks + (e & 3) + 8 * (e >> 2)) * DC_T * 4, 0);
        }
        af[ks] = dc_frag8(v);
      }
#pragma unroll
      for (int mt = 0; mt < K::MT; ++mt) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < K::NK2; ++ks)
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, s.w2f[(mt * K::NK2 + ks) * 64 + lane]), af[ks], acc, 0, 0, 0);
        const float* bp = s.pz[0] + mt * 16 + hh * 8;
        const float* bq = s.pz[1] + mt * 16 + hh * 8;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float p = acc[r] + bp[r], q = acc[r + 8] + bq[r];
          acc[r] = p; acc[r + 8] = q;
          t1 += p + q; t2 = fmaf(p, p, fmaf(q, q, t2));
          if (SAVE) {
            __builtin_amdgcn_raw_buffer_store_b16((short)rfx_bf16_bits(p), zrs, (voff0 >> 1) + 64 * ct, DC_ROFF(mt, r, 2), 0);
            __builtin_amdgcn_raw_buffer_store_b16((short)rfx_bf16_bits(q), zrs, (voff0 >> 1) + 64 * ct, DC_ROFF(mt, r, 2) + C * DC_T * 2, 0);
          }
        }
        z[ct][mt] = acc;
      }
    }
    dc_wg_sum2(s.red, wave, lane, t1, t2, 2);
    const float mu2 = t1 * inv2, rs2 = rsqrtf(fmaxf(t2 * inv2 - mu2 * mu2, 0.f) + a.eps);
    if (SAVE && tid == 0) {
      a.stats[n] = mu1; a.stats[a.N + n] = rs1; a.stats[2 * (int64_t)a.N + n] = mu2; a.stats[3 * (int64_t)a.N + n] = rs2;
    }
    // ---- out = x + scale * gn(z)_value * sigmoid(gn(z)_gate)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
      for (int mt = 0; mt < K::MT; ++mt) {
        const int o = mt * 16 + hh * 8;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float ap_ = rs2 * s.pz[2][o + r], aq_ = rs2 * s.pz[3][o + r];
          const float zp = fmaf(z[ct][mt][r] - mu2, ap_, s.pz[4][o + r]);
          const float zq = fmaf(z[ct][mt][r + 8] - mu2, aq_, s.pz[5][o + r]);
          const float u = zp * dc_sigmoid(zq);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, fmaf(s.pz[6][o + r], u, xr[ct][mt][r])), ors,
                                                voff0 + 128 * ct, DC_ROFF(mt, r, 4), 0);
        }
      }
    }
  }
}

template <int C, bool SAVE>
static int dconv_launch_fwd(const DconvArgs& a, hipStream_t s) {
  const size_t lds = sizeof(DcLds<C>);
  if (lds > 80 * 1024) return -1;                       // two workgroups per CU
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(dconv_fwd_kernel<C, SAVE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds) != hipSuccess) return -3;
  const int grid = a.N < 512 ? a.N : 512;
  hipLaunchKernelGGL((dconv_fwd_kernel<C, SAVE>), dim3(grid), dim3(256), lds, s, a);
  RFX_CHECK_LAUNCH();
  return 0;
}

extern "C" int rfx_dconv_layer_ok(int32_t C, int32_t T, int32_t dil) { return T == DC_T && C == 48 && (dil == 1 || dil == 2); }

extern "C" int rfx_dconv_layer_fwd(const float* x, float* out, int32_t N, int32_t C, int32_t T, int32_t dil, const float* w1,
                                   const float* b1, const float* gn1w, const float* gn1b, const float* w2, const float* b2,
                                   const float* gn2w, const float* gn2b, const float* scale, float eps, void* h16, void* z16,
                                   float* a_out, float* stats, void* stream) {
  DconvArgs a{};
  a.x = x; a.out = out; a.w1 = w1; a.b1 = b1; a.gn1w = gn1w; a.gn1b = gn1b; a.w2 = w2; a.b2 = b2; a.gn2w = gn2w; a.gn2b = gn2b;
  a.scale = scale; a.N = N; a.dil = dil; a.eps = eps;
  a.h16 = (uint16_t*)h16; a.z16 = (uint16_t*)z16; a.a_out = a_out; a.stats = stats;
  if (!(x && out && w1 && b1 && gn1w && gn1b && w2 && b2 && gn2w && gn2b && scale && N > 0) || !rfx_dconv_layer_ok(C, T, dil)) return -1;
  const bool save = h16 || z16 || a_out || stats;
  if (save && !(h16 && z16 && a_out && stats)) return -1;
  return save ? dconv_launch_fwd<48, true>(a, (hipStream_t)stream) : dconv_launch_fwd<48, false>(a, (hipStream_t)stream);
}
