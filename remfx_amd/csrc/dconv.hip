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
  // backward (dconv_bwd_kernel): upstream gradient in, operands of the two weight-gradient GEMMs + parameter-gradient partial sums out
  const float* g;        // (N, C, T)
  uint16_t* dz;          // (N, 2C, T) bf16
  uint16_t* dh;          // (N, H, T) bf16
  float* partial;        // (gridDim.x, 5C + 2H)
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


// ---------------------------------------------------------------------------------------------------------------------------------
// Fused BACKWARD of the depth-layer (round 4).  Nothing but the layer input x is needed: the forward is recomputed from it, the
// gradient is walked back in registers and one pass over (x, gy) produces
//     dx                       (N, C, T) fp32
//     dz, dh (bf16), a (fp32)  the operands of the two weight-gradient GEMMs (rfx_gemm_wgrad: dW2 = dz a^T, dW1 = dh * x), and
//     per-workgroup partial sums of the LayerScale / GroupNorm affine gradients.
// It replaces, per layer, GroupNorm+GLU+LayerScale backward, the 1x1 input-gradient GEMM, GroupNorm+GELU backward and the 3-tap
// input-gradient GEMM (4 launches, 2.9 ms at N = 32768 in the r04 sequence profile) and lets the forward run without storing h / z.
//
// Round 3 built this once and dropped it (one 4-wave workgroup per CU, 152 per-lane accumulators for the parameter gradients,
// 250 live values per lane -> scratch in the sample loop, 9.4 ms per layer).  What is different here:
//   * a workgroup is 16 waves = TWO samples, a wave owns ONE 32-position tile: 128 registers per lane, four waves per SIMD, so the
//     latency of every phase of one wave is covered by three others;
//   * the small parameter gradients are row sums over positions: every lane adds its element into an LDS accumulator
//     [row][32 position lanes] with ds_add_f32 (no return, conflict-free: a half-wave = 32 consecutive banks); the rows are summed
//     over the 32 lanes once, when the workgroup retires;
//   * x of the next sample pair is fetched into the registers the staged image freed, gy before the first GEMM, and gy again
//     (L2) for the final residual add -- no value is carried across more than two phases.
// Six workgroup barriers per sample pair (image, three pairs of sample-wide sums, dh image).
template <int C>
struct DcLdsBwd {
  using K = DcCfg<C>;
  static constexpr int HP = 16 * K::NK2, DHP = HP + 8;              // dh image: hidden rows padded to whole K steps + 8 (48-byte positions)
  static constexpr int MTX = (C + 31) / 32;                         // 32-row tiles over C (input-gradient GEMM)
  static constexpr int ROWS = 5 * C + 2 * K::H;                     // dscale C | dgn2w 2C | dgn2b 2C | dgn1w H | dgn1b H
  uint16_t xs[2][(DC_T + 2 * DC_PADR) * K::CP];                     // channels-last bf16 image of x, one per sample
  uint16_t dhs[2][(DC_T + 2 * DC_PADR) * DHP];                      // channels-last bf16 image of dh
  uint4 w1f[K::NK1 * 64];                                           // A fragments: W1 (forward GEMM1)
  uint4 w2f[K::MT * K::NK2 * 64];                                   //              row-permuted W2 (forward GEMM2)
  uint4 w2tf[K::MT * 2 * 64];                                       //              W2^T per z tile: K step 0 = value rows, 1 = gate rows
  uint4 w1tf[MTX * 3 * K::NK2 * 64];                                //              W1^T per 32-channel tile, K steps (tap, 16 hidden)
  float b1[32], g1[32], be1[32];
  float pz[7][K::MT * 16];
  float accf[ROWS * 32];
  float red[2][16][2];
};

template <int C>
__global__ __launch_bounds__(1024, 4) void dconv_bwd_kernel(const DconvArgs a) {
  using K = DcCfg<C>;
  using L = DcLdsBwd<C>;
  extern __shared__ __attribute__((aligned(16))) unsigned char dc_smem[];
  L& s = *reinterpret_cast<L*>(dc_smem);
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, hh = lane >> 5;
  // wave-uniform BY CONSTRUCTION for the compiler too (readfirstlane): everything derived from the wave index -- the sample of the pair,
  // hence every buffer descriptor -- must live in SGPRs, or each buffer access becomes a waterfall loop over descriptor VGPRs
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sidx = wave >> 3, pt = wave & 7, t = 32 * pt + j;       // sample of the pair, position tile, position
  // ---- parameter tables (once per workgroup)
  for (int idx = tid; idx < K::NK1 * 64; idx += 1024) {
    const int ks = idx >> 6, ln = idx & 63, i = ln & 31, h2 = ln >> 5;
    const int g = 2 * ks + h2, tap = g / K::G8, cg = g % K::G8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = i < K::H ? a.w1[((int64_t)i * C + cg * 8 + e) * 3 + tap] : 0.f;
    s.w1f[idx] = __builtin_bit_cast(uint4, dc_frag8(v));
  }
  for (int idx = tid; idx < K::MT * K::NK2 * 64; idx += 1024) {
    const int ln = idx & 63, ks = (idx >> 6) % K::NK2, mt = (idx >> 6) / K::NK2, i = ln & 31, h2 = ln >> 5;
    const int row = i < 16 ? 16 * mt + i : C + 16 * mt + (i - 16);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 16 * ks + (e & 3) + 8 * (e >> 2) + 4 * h2;
      v[e] = k < K::H ? a.w2[(int64_t)row * K::H + k] : 0.f;
    }
    s.w2f[idx] = __builtin_bit_cast(uint4, dc_frag8(v));
  }
  for (int idx = tid; idx < K::MT * 2 * 64; idx += 1024) {
    const int ln = idx & 63, sq = (idx >> 6) & 1, mt = idx >> 7, i = ln & 31, h2 = ln >> 5;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = 16 * mt + (e & 3) + 8 * (e >> 2) + 4 * h2;
      const int row = sq ? C + c : c;
      v[e] = i < K::H ? a.w2[(int64_t)row * K::H + i] : 0.f;
    }
    s.w2tf[idx] = __builtin_bit_cast(uint4, dc_frag8(v));
  }
  for (int idx = tid; idx < L::MTX * 3 * K::NK2 * 64; idx += 1024) {
    const int ln = idx & 63, q = (idx >> 6) % (3 * K::NK2), mtx = (idx >> 6) / (3 * K::NK2), i = ln & 31, h2 = ln >> 5;
    const int tap = q / K::NK2, ks = q % K::NK2, c = 32 * mtx + i;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int m = 16 * ks + 8 * h2 + e;
      v[e] = (c < C && m < K::H) ? a.w1[((int64_t)m * C + c) * 3 + tap] : 0.f;
    }
    s.w1tf[idx] = __builtin_bit_cast(uint4, dc_frag8(v));
  }
  if (tid < 32) {
    s.b1[tid] = tid < K::H ? a.b1[tid] : 0.f;
    s.g1[tid] = tid < K::H ? a.gn1w[tid] : 0.f;
    s.be1[tid] = tid < K::H ? a.gn1b[tid] : 0.f;
  }
  for (int idx = tid; idx < K::MT * 16; idx += 1024) {
    const int mt = idx >> 4, h2 = (idx >> 3) & 1, r = idx & 7, c = dc_chan(mt, r, h2);
    s.pz[0][idx] = a.b2[c];     s.pz[1][idx] = a.b2[C + c];
    s.pz[2][idx] = a.gn2w[c];   s.pz[3][idx] = a.gn2w[C + c];
    s.pz[4][idx] = a.gn2b[c];   s.pz[5][idx] = a.gn2b[C + c];
    s.pz[6][idx] = a.scale[c];
  }
  for (int idx = tid; idx < 2 * (DC_T + 2 * DC_PADR) * K::CP; idx += 1024) (&s.xs[0][0])[idx] = 0;      // zero rows / padding columns stay zero
  for (int idx = tid; idx < 2 * (DC_T + 2 * DC_PADR) * L::DHP; idx += 1024) (&s.dhs[0][0])[idx] = 0;
  for (int idx = tid; idx < L::ROWS * 32; idx += 1024) s.accf[idx] = 0.f;
  __syncthreads();              // the zero fills above touch rows other waves are about to write (S0 of the first pair)

  const uint32_t voff0 = (uint32_t)((4 * hh * DC_T + t) * 4);      // byte offset of (row 4 hh, position t) in an fp32 (rows, T) sample
  const float inv1 = 1.0f / (float)(K::H * DC_T), inv2 = 1.0f / (float)(2 * C * DC_T);
  uint16_t* xsi = s.xs[sidx];
  uint16_t* dhi = s.dhs[sidx];
  const int npairs = (a.N + 1) >> 1;
  float xr[K::MT][8];
  {
    const int n0 = 2 * (int)blockIdx.x + sidx;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + (int64_t)(n0 < a.N ? n0 : 0) * C * DC_T), 0,
                                                                         n0 < a.N ? C * DC_T * 4 : 0, 0x00020000);
    dc_load_x<C>(xrs, voff0, xr);
  }
  for (int p = blockIdx.x; p < npairs; p += gridDim.x) {
    const int n = 2 * p + sidx;
    const bool live = n < a.N;                                      // odd N: the second sample of the last pair only keeps the barriers company
    const int64_t nn = live ? n : 0;
    const int rec = live ? 1 : 0;
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.g + nn * C * DC_T), 0, rec * C * DC_T * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(a.out + nn * C * DC_T, 0, rec * C * DC_T * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(a.a_out + nn * K::H * DC_T, 0, rec * K::H * DC_T * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t zrs = __builtin_amdgcn_make_buffer_rsrc(a.dz + nn * 2 * C * DC_T, 0, rec * 2 * C * DC_T * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(a.dh + nn * K::H * DC_T, 0, rec * K::H * DC_T * 2, 0x00020000);
    // ---- S0: channels-last bf16 image of x
    {
      uint16_t* row = xsi + (t + DC_PADR) * K::CP;
#pragma unroll
      for (int mt = 0; mt < K::MT; ++mt)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          *reinterpret_cast<uint2*>(row + 16 * mt + 8 * q + 4 * hh) =
              make_uint2(dc_pack2(xr[mt][4 * q], xr[mt][4 * q + 1]), dc_pack2(xr[mt][4 * q + 2], xr[mt][4 * q + 3]));
    }
    __syncthreads();                                                // B1
    // ---- S1: gy on its way; GEMM1 (recompute h) + GroupNorm(1, H) statistics
    float gr[K::MT][8];
    dc_load_x<C>(grs, voff0, gr);
    f32x16 hacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) hacc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < K::NK1; ++ks) {
      const int tap = (2 * ks) / K::G8, cg = (2 * ks) % K::G8;
      const uint16_t* q = xsi + (t + DC_PADR + (tap - 1) * a.dil) * K::CP + (cg + hh) * 8;
      hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, s.w1f[ks * 64 + lane]),
                                                     __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(q)), hacc, 0, 0, 0);
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = hacc[r] + s.b1[dc_hrow(r, hh)];                // rows beyond H: zero weight rows + zero bias = exactly 0
      hacc[r] = v;
      s1 += v; s2 = fmaf(v, v, s2);
    }
    s1 = rfx_wave_sum(s1); s2 = rfx_wave_sum(s2);
    if (lane == 0) { s.red[0][wave][0] = s1; s.red[0][wave][1] = s2; }
    __syncthreads();                                                // B2
    s1 = 0.f; s2 = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) { s1 += s.red[0][8 * sidx + w][0]; s2 += s.red[0][8 * sidx + w][1]; }
    const float mu1 = s1 * inv1, rs1 = rsqrtf(fmaxf(s2 * inv1 - mu1 * mu1, 0.f) + a.eps);
    // ---- S2: a = GELU(gn(h)) (stored for the dW2 GEMM) -> GEMM2 from registers -> statistics of z.  hacc becomes the normalised h.
    f32x16 z[K::MT];
    float t1 = 0.f, t2 = 0.f;
    {
      bf16x8 af[K::NK2];
#pragma unroll
      for (int ks = 0; ks < K::NK2; ++ks) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int m = 16 * ks + dc_hrow(e, hh);
          const float hn_ = (hacc[8 * ks + e] - mu1) * rs1;
          hacc[8 * ks + e] = hn_;
          v[e] = dc_gelu(fmaf(hn_, s.g1[m], s.be1[m]));             // rows beyond H: zero affine -> GELU(0) = 0
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v[e]), ars, voff0, (16 * ks + (e & 3) + 8 * (e >> 2)) * DC_T * 4, 0);
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
          const float pv = acc[r] + bp[r], qv = acc[r + 8] + bq[r];
          acc[r] = pv; acc[r + 8] = qv;
          t1 += pv + qv; t2 = fmaf(pv, pv, fmaf(qv, qv, t2));
        }
        z[mt] = acc;
      }
    }
    t1 = rfx_wave_sum(t1); t2 = rfx_wave_sum(t2);
    if (lane == 0) { s.red[1][wave][0] = t1; s.red[1][wave][1] = t2; }
    __syncthreads();                                                // B3
    t1 = 0.f; t2 = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) { t1 += s.red[1][8 * sidx + w][0]; t2 += s.red[1][8 * sidx + w][1]; }
    const float mu2 = t1 * inv2, rs2 = rsqrtf(fmaxf(t2 * inv2 - mu2 * mu2, 0.f) + a.eps);
    // ---- S3 (phase A): GLU + LayerScale backward, parameter-gradient row sums, the two sample-wide sums of GroupNorm(1, 2C).
    // Kept for phase B as bf16 pairs (they only ever feed bf16 GEMM operands): normalised z (value, gate), gamma * dzn (value, gate).
    uint32_t zh_pk[K::MT][8], e_pk[K::MT][8];
    float S1 = 0.f, S2 = 0.f;
    float* accj = s.accf + j;
#pragma unroll
    for (int mt = 0; mt < K::MT; ++mt) {
      const int o = mt * 16 + hh * 8;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int c = dc_chan(mt, r, hh);
        const float gp = s.pz[2][o + r], gq = s.pz[3][o + r];
        const float zhp = (z[mt][r] - mu2) * rs2, zhq = (z[mt][r + 8] - mu2) * rs2;
        const float znp = fmaf(zhp, gp, s.pz[4][o + r]), znq = fmaf(zhq, gq, s.pz[5][o + r]);
        const float sg = dc_sigmoid(znq);
        const float g = live ? gr[mt][r] : 0.f;
        const float du = g * s.pz[6][o + r];
        const float dp = du * sg, dq = du * znp * sg * (1.0f - sg);
        atomicAdd(accj + 32 * c, g * znp * sg);                                  // dscale[c]
        atomicAdd(accj + 32 * (C + c), dp * zhp);                                // dgn2w value / gate rows
        atomicAdd(accj + 32 * (2 * C + c), dq * zhq);
        atomicAdd(accj + 32 * (3 * C + c), dp);                                  // dgn2b
        atomicAdd(accj + 32 * (4 * C + c), dq);
        const float ep = gp * dp, eq = gq * dq;
        S1 += ep + eq;
        S2 = fmaf(ep, zhp, fmaf(eq, zhq, S2));
        zh_pk[mt][r] = dc_pack2(zhp, zhq);
        e_pk[mt][r] = dc_pack2(ep, eq);
      }
    }
    S1 = rfx_wave_sum(S1); S2 = rfx_wave_sum(S2);
    if (lane == 0) { s.red[0][wave][0] = S1; s.red[0][wave][1] = S2; }
    __syncthreads();                                                // B4
    S1 = 0.f; S2 = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) { S1 += s.red[0][8 * sidx + w][0]; S2 += s.red[0][8 * sidx + w][1]; }
    const float m1 = S1 * inv2, m2 = S2 * inv2;
    // ---- S4 (phase B): dz -> global (bf16, pairs of positions per store) and -> da = W2^T dz;  GELU', GroupNorm(1, H) backward sums
    f32x16 da;
#pragma unroll
    for (int r = 0; r < 16; ++r) da[r] = 0.f;
    {
      const bool odd = j & 1;
      const uint32_t sel = odd ? 0x03020706u : 0x05040100u;
      // even lanes store the value row of positions (t, t + 1), odd lanes the gate row of (t - 1, t)
      const uint32_t vz = (uint32_t)((4 * hh * DC_T + t) * 2) + (odd ? (uint32_t)(C * DC_T * 2) - 2u : 0u);
#pragma unroll
      for (int mt = 0; mt < K::MT; ++mt) {
        float vp[8], vq[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const uint32_t zk = zh_pk[mt][r], ek = e_pk[mt][r];
          vp[r] = rs2 * (__uint_as_float(ek << 16) - m1 - __uint_as_float(zk << 16) * m2);
          vq[r] = rs2 * (__uint_as_float(ek & 0xffff0000u) - m1 - __uint_as_float(zk & 0xffff0000u) * m2);
          const uint32_t w = rfx_cvt_pk_bf16(vp[r], vq[r]);
          const uint32_t recv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0xB1, 0xF, 0xF, false);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_amdgcn_perm(recv, w, sel), zrs, vz, DC_ROFF(mt, r, 2), 0);
        }
        da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, s.w2tf[(mt * 2 + 0) * 64 + lane]), dc_frag8(vp), da, 0, 0, 0);
        da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, s.w2tf[(mt * 2 + 1) * 64 + lane]), dc_frag8(vq), da, 0, 0, 0);
      }
    }
    // the next pair's x: its registers have been free since S0
    {
      const int n2 = 2 * (p + (int)gridDim.x) + sidx;
      const bool l2 = n2 < a.N;
      const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + (int64_t)(l2 ? n2 : 0) * C * DC_T), 0,
                                                                           l2 ? C * DC_T * 4 : 0, 0x00020000);
      dc_load_x<C>(xrs, voff0, xr);
    }
    float Q1 = 0.f, Q2 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = dc_hrow(r, hh);
      const float hn_ = hacc[r];
      const float dhn = m < K::H ? da[r] * dc_gelu_grad(fmaf(hn_, s.g1[m], s.be1[m])) : 0.f;
      if (m < K::H) {                                                // compile-time per (r) up to the lane half: rows beyond H have no slot
        atomicAdd(accj + 32 * (5 * C + m), dhn * hn_);                // dgn1w
        atomicAdd(accj + 32 * (5 * C + K::H + m), dhn);               // dgn1b
      }
      const float e = s.g1[m] * dhn;
      Q1 += e; Q2 = fmaf(e, hn_, Q2);
      da[r] = e;
    }
    Q1 = rfx_wave_sum(Q1); Q2 = rfx_wave_sum(Q2);
    if (lane == 0) { s.red[1][wave][0] = Q1; s.red[1][wave][1] = Q2; }
    __syncthreads();                                                // B5
    Q1 = 0.f; Q2 = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) { Q1 += s.red[1][8 * sidx + w][0]; Q2 += s.red[1][8 * sidx + w][1]; }
    const float q1 = Q1 * inv1, q2 = Q2 * inv1;
    // ---- S5: dh -> its channels-last image (taps of the input-gradient GEMM) and -> global (bf16, for dW1); gy again for the residual
    {
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = dc_hrow(r, hh);
        v[r] = m < K::H ? rs1 * (da[r] - q1 - hacc[r] * q2) : 0.f;
        // rows beyond H fall outside the (H, T) buffer: dropped by the range check
        __builtin_amdgcn_raw_buffer_store_b16((short)rfx_bf16_bits(v[r]), hrs, (voff0 >> 1), ((r & 3) + 8 * (r >> 2)) * DC_T * 2, 0);
      }
      uint16_t* row = dhi + (t + DC_PADR) * L::DHP;
#pragma unroll
      for (int q = 0; q < 2 * K::NK2; ++q)       // registers 4q .. 4q + 3 = hidden rows 8q + 4hh .. + 3
        *reinterpret_cast<uint2*>(row + 8 * q + 4 * hh) = make_uint2(dc_pack2(v[4 * q], v[4 * q + 1]), dc_pack2(v[4 * q + 2], v[4 * q + 3]));
    }
    dc_load_x<C>(grs, voff0, gr);
    __syncthreads();                                                // B6
    // ---- S6: dx = gy + W1^T * dh
#pragma unroll
    for (int mtx = 0; mtx < L::MTX; ++mtx) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int tap = 0; tap < 3; ++tap)
#pragma unroll
        for (int ks = 0; ks < K::NK2; ++ks) {
          const uint16_t* q = dhi + (t + DC_PADR - (tap - 1) * a.dil) * L::DHP + 16 * ks + 8 * hh;
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, s.w1tf[((mtx * 3 + tap) * K::NK2 + ks) * 64 + lane]),
                                                        __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(q)), acc, 0, 0, 0);
        }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int mt = 2 * mtx + (r >> 3);
        if (mt < K::MT)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, gr[mt][r & 7] + acc[r]), ors, voff0, DC_ROFF(mt, r & 7, 4), 0);
      }
    }
  }
  // ---- row sums of the accumulators over the 32 position lanes: one row of `partial` per workgroup
  __syncthreads();
  float* outp = a.partial + (int64_t)blockIdx.x * L::ROWS;
  for (int row = tid; row < L::ROWS; row += 1024) {
    float v = 0.f;
#pragma unroll 8
    for (int q = 0; q < 32; ++q) v += s.accf[row * 32 + ((q + row) & 31)];          // skewed start: rows of one wave hit different banks
    outp[row] = v;
  }
}

template <int C>
static int dconv_launch_bwd(const DconvArgs& a, int grid, hipStream_t s) {
  const size_t lds = sizeof(DcLdsBwd<C>);
  if (lds > 160 * 1024) return -1;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(dconv_bwd_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess) return -3;
    attr_set = true;
  }
  hipLaunchKernelGGL(dconv_bwd_kernel<C>, dim3(grid), dim3(1024), lds, s, a);
  RFX_CHECK_LAUNCH();
  return 0;
}

// partial: rfx_dconv_layer_bwd_rows(N) rows of 5C + 2H floats [dscale C | dgn2w 2C | dgn2b 2C | dgn1w H | dgn1b H]; the caller sums the rows
extern "C" int rfx_dconv_layer_ok(int32_t C, int32_t T, int32_t dil);
extern "C" int rfx_dconv_layer_bwd_rows(int32_t N) { const int pairs = (N + 1) / 2; return pairs < 256 ? pairs : 256; }
extern "C" int rfx_dconv_layer_bwd(const float* x, const float* g, float* gx, int32_t N, int32_t C, int32_t T, int32_t dil,
                                   const float* w1, const float* b1, const float* gn1w, const float* gn1b, const float* w2,
                                   const float* b2, const float* gn2w, const float* gn2b, const float* scale, float eps,
                                   void* dz_bf16, float* a_out, void* dh_bf16, float* partial, void* stream) {
  DconvArgs a{};
  a.x = x; a.g = g; a.out = gx; a.w1 = w1; a.b1 = b1; a.gn1w = gn1w; a.gn1b = gn1b; a.w2 = w2; a.b2 = b2; a.gn2w = gn2w; a.gn2b = gn2b;
  a.scale = scale; a.N = N; a.dil = dil; a.eps = eps;
  a.dz = (uint16_t*)dz_bf16; a.a_out = a_out; a.dh = (uint16_t*)dh_bf16; a.partial = partial;
  if (!(x && g && gx && w1 && b1 && gn1w && gn1b && w2 && b2 && gn2w && gn2b && scale && dz_bf16 && a_out && dh_bf16 && partial && N > 0) ||
      !rfx_dconv_layer_ok(C, T, dil)) return -1;
  return dconv_launch_bwd<48>(a, rfx_dconv_layer_bwd_rows(N), (hipStream_t)stream);
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
