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
// The backward kernel recomputes the forward from x (nothing but the layer input is saved), accumulates the small
// parameter gradients (LayerScale, both GroupNorm affines) in per-lane registers across its samples, and emits dz, a and
// dh for the two weight-gradient GEMMs.
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define DC_T 256
#define DC_PADR 2      // zero rows either side of the channels-last image (largest dilation of the two depth layers)

struct DconvArgs {
  const float* x;        // (N, C, T)
  float* out;            // fwd: (N, C, T);  bwd: gx
  const float* g;        // bwd: upstream gradient (N, C, T)
  const float *w1, *b1, *gn1w, *gn1b, *w2, *b2, *gn2w, *gn2b, *scale;
  // bwd outputs for the weight-gradient GEMMs
  uint16_t* dz;          // (N, 2C, T) bf16
  float* a_out;          // (N, H, T)
  uint16_t* dh;          // (N, H, T) bf16
  float* partial;        // (gridDim.x, NPART) per-workgroup sums of the small parameter gradients
  int N, dil;
  float eps;
};

template <int C>
struct DcCfg {
  static constexpr int H = C / 4, G8 = C / 8, NK1 = 3 * G8 / 2, MT = C / 16, NK2 = (H + 15) / 16, CP = C + 8;
  static constexpr int MTX = C / 32 + (C % 32 ? 1 : 0);     // 32-row tiles over C (input-gradient GEMM)
  static constexpr int HP = 16 * NK2;                       // hidden rows padded to whole K steps
  // per-workgroup partial sums: dscale[C], dgn2w[2C], dgn2b[2C], dgn1w[H], dgn1b[H]
  static constexpr int NPART = C + 4 * C + 2 * H;
};

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

// x registers of one 32-column tile in the (mt, r) layout + the channels-last bf16 image
template <int C>
__device__ __forceinline__ void dc_load_x(const float* __restrict__ xp, int t, int hh, float (&xr)[DcCfg<C>::MT][8]) {
#pragma unroll
  for (int mt = 0; mt < DcCfg<C>::MT; ++mt)
#pragma unroll
    for (int r = 0; r < 8; ++r) xr[mt][r] = xp[(int64_t)dc_chan(mt, r, hh) * DC_T + t];
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

template <int C>
__global__ __launch_bounds__(256, 1) void dconv_fwd_kernel(const DconvArgs a) {
  using K = DcCfg<C>;
  extern __shared__ __attribute__((aligned(16))) unsigned char dc_smem[];
  DcLds<C>& s = *reinterpret_cast<DcLds<C>*>(dc_smem);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, hh = lane >> 5;
  dc_load_params<C>(s, a, tid);
  float xr[2][K::MT][8];
  int n = blockIdx.x;
  if (n < a.N) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) dc_load_x<C>(a.x + (int64_t)n * C * DC_T, 64 * wave + 32 * ct + j, hh, xr[ct]);
  }
  for (; n < a.N; n += gridDim.x) {
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
        if (m < K::H) { s1 += v; s2 = fmaf(v, v, s2); }
      }
    }
    dc_wg_sum2(s.red, wave, lane, s1, s2, 0);
    const float inv1 = 1.0f / (float)(K::H * DC_T);
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
          v[e] = m < K::H ? rfx_gelu(hn) : 0.f;
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
        }
        z[ct][mt] = acc;
      }
    }
    dc_wg_sum2(s.red, wave, lane, t1, t2, 2);
    const float inv2 = 1.0f / (float)(2 * C * DC_T);
    const float mu2 = t1 * inv2, rs2 = rsqrtf(fmaxf(t2 * inv2 - mu2 * mu2, 0.f) + a.eps);
    // ---- prefetch the next sample's x while this one is finished (the current x is consumed below)
    float xn[2][K::MT][8];
    const int nn = n + gridDim.x;
    if (nn < a.N) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) dc_load_x<C>(a.x + (int64_t)nn * C * DC_T, 64 * wave + 32 * ct + j, hh, xn[ct]);
    }
    // ---- out = x + scale * gn(z)_value * sigmoid(gn(z)_gate)
    float* op = a.out + (int64_t)n * C * DC_T;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int t = 64 * wave + 32 * ct + j;
#pragma unroll
      for (int mt = 0; mt < K::MT; ++mt) {
        const int o = mt * 16 + hh * 8;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float zp = (z[ct][mt][r] - mu2) * rs2 * s.pz[2][o + r] + s.pz[4][o + r];
          const float zq = (z[ct][mt][r + 8] - mu2) * rs2 * s.pz[3][o + r] + s.pz[5][o + r];
          const float u = zp * rfx_sigmoid(zq);
          op[(int64_t)dc_chan(mt, r, hh) * DC_T + t] = fmaf(s.pz[6][o + r], u, xr[ct][mt][r]);
        }
      }
    }
    if (nn < a.N) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int mt = 0; mt < K::MT; ++mt)
#pragma unroll
          for (int r = 0; r < 8; ++r) xr[ct][mt][r] = xn[ct][mt][r];
    }
  }
}

// ---- backward --------------------------------------------------------------------------------------------------------------
// Recomputes h, a, z from the layer input, then walks the chain back:
//   dzn (GLU + LayerScale backward) -> GroupNorm(1, 2C) backward -> dz -> da = W2^T dz (MFMA, B operand = dz registers)
//   -> GELU' -> GroupNorm(1, H) backward -> dh -> LDS (channels-last) -> dx = g + W1^T * dh over the three taps (MFMA).
// dz (bf16), a (fp32) and dh (bf16) go to global memory for the two weight-gradient GEMMs (dW2 = dz a^T, dW1 = dh x^T run on
// the existing wgrad kernels, which also produce the conv biases); the LayerScale and GroupNorm affine gradients are summed
// in per-lane registers over all samples of the workgroup and reduced once at the end into partial[blockIdx.x][.]
// (deterministic: the host adds the rows in order).
template <int C>
struct DcLdsB : DcLds<C> {
  using K = DcCfg<C>;
  static constexpr int DHP = K::HP + 8;
  uint4 w2tf[K::MT * 2 * 64];                       // A fragments of W2^T per z tile: K step 0 = value rows, 1 = gate rows
  uint4 w1tf[K::MTX * 3 * K::NK2 * 64];             // A fragments of W1^T per 32-channel tile, K steps (tap, 16 hidden)
  uint16_t dhs[(DC_T + 2 * DC_PADR) * DHP];         // channels-last bf16 image of dh, zero rows either side
  static constexpr int NA = K::MT * 40 + 32;        // per-lane accumulators: ds MT*8, gn2w / gn2b MT*16 each, gn1w / gn1b 16 each
  float acc[4][2][NA];                              // end-of-kernel reduction scratch: [wave][hh][accumulator]
};

template <int C>
__device__ __forceinline__ void dc_load_params_bwd(DcLdsB<C>& s, const DconvArgs& a, int tid) {
  using K = DcCfg<C>;
  for (int idx = tid; idx < K::MT * 2 * 64; idx += 256) {
    const int lane = idx & 63, sq = (idx >> 6) & 1, mt = idx >> 7, i = lane & 31, hh = lane >> 5;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = 16 * mt + (e & 3) + 8 * (e >> 2) + 4 * hh;
      const int row = sq ? C + c : c;
      v[e] = i < K::H ? a.w2[(int64_t)row * K::H + i] : 0.f;
    }
    s.w2tf[idx] = __builtin_bit_cast(uint4, dc_frag8(v));
  }
  for (int idx = tid; idx < K::MTX * 3 * K::NK2 * 64; idx += 256) {
    const int lane = idx & 63, q = (idx >> 6) % (3 * K::NK2), mtx = (idx >> 6) / (3 * K::NK2), i = lane & 31, hh = lane >> 5;
    const int tap = q / K::NK2, ks = q % K::NK2, c = 32 * mtx + i;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int m = 16 * ks + 8 * hh + e;
      v[e] = (c < C && m < K::H) ? a.w1[((int64_t)m * C + c) * 3 + tap] : 0.f;
    }
    s.w1tf[idx] = __builtin_bit_cast(uint4, dc_frag8(v));
  }
  for (int idx = tid; idx < (DC_T + 2 * DC_PADR) * DcLdsB<C>::DHP; idx += 256) s.dhs[idx] = 0;     // pad rows / columns stay zero
}

template <int C>
__global__ __launch_bounds__(256, 1) void dconv_bwd_kernel(const DconvArgs a) {
  using K = DcCfg<C>;
  using L = DcLdsB<C>;
  extern __shared__ __attribute__((aligned(16))) unsigned char dc_smem[];
  L& s = *reinterpret_cast<L*>(dc_smem);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, hh = lane >> 5;
  dc_load_params<C>(s, a, tid);
  dc_load_params_bwd<C>(s, a, tid);
  // per-lane accumulators over all samples / both column tiles of this wave
  float ds_acc[K::MT][8], g2w_acc[K::MT][16], g2b_acc[K::MT][16], g1w_acc[16], g1b_acc[16];
#pragma unroll
  for (int mt = 0; mt < K::MT; ++mt) {
#pragma unroll
    for (int r = 0; r < 8; ++r) ds_acc[mt][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { g2w_acc[mt][r] = 0.f; g2b_acc[mt][r] = 0.f; }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) { g1w_acc[r] = 0.f; g1b_acc[r] = 0.f; }
  float xr[2][K::MT][8], gr[2][K::MT][8];
  int n = blockIdx.x;
  if (n < a.N) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      dc_load_x<C>(a.x + (int64_t)n * C * DC_T, 64 * wave + 32 * ct + j, hh, xr[ct]);
      dc_load_x<C>(a.g + (int64_t)n * C * DC_T, 64 * wave + 32 * ct + j, hh, gr[ct]);
    }
  }
  const float inv1 = 1.0f / (float)(K::H * DC_T), inv2 = 1.0f / (float)(2 * C * DC_T);
  for (; n < a.N; n += gridDim.x) {
    __syncthreads();
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) dc_store_xs<C>(s, 64 * wave + 32 * ct + j, hh, xr[ct]);
    __syncthreads();
    const int nn = n + gridDim.x;
    if (nn < a.N) {                       // x is dead once its image is in LDS: fetch the next sample's
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) dc_load_x<C>(a.x + (int64_t)nn * C * DC_T, 64 * wave + 32 * ct + j, hh, xr[ct]);
    }
    // ---- recompute h and its statistics
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
        if (m < K::H) { s1 += v; s2 = fmaf(v, v, s2); }
      }
    }
    dc_wg_sum2(s.red, wave, lane, s1, s2, 0);
    const float mu1 = s1 * inv1, rs1 = rsqrtf(fmaxf(s2 * inv1 - mu1 * mu1, 0.f) + a.eps);
    // ---- a, GEMM2, statistics of z
    f32x16 z[2][K::MT];
    float t1 = 0.f, t2 = 0.f;
    float* ap = a.a_out + (int64_t)n * K::H * DC_T;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int t = 64 * wave + 32 * ct + j;
      bf16x8 af[K::NK2];
#pragma unroll
      for (int ks = 0; ks < K::NK2; ++ks) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int m = 16 * ks + dc_hrow(e, hh);
          const float hn = (hacc[ct][8 * ks + e] - mu1) * rs1 * s.g1[m] + s.be1[m];
          v[e] = m < K::H ? rfx_gelu(hn) : 0.f;
          if (m < K::H) ap[(int64_t)m * DC_T + t] = v[e];
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
        }
        z[ct][mt] = acc;
      }
    }
    dc_wg_sum2(s.red, wave, lane, t1, t2, 2);
    const float mu2 = t1 * inv2, rs2 = rsqrtf(fmaxf(t2 * inv2 - mu2 * mu2, 0.f) + a.eps);
    // ---- phase A: GLU + LayerScale backward, parameter-gradient accumulation, the two sample-wide sums of GroupNorm(1, 2C)
    uint32_t dzn[2][K::MT][8];              // (dzn_value, dzn_gate) as a bf16 pair: they only ever feed GEMM operands
    float S1 = 0.f, S2 = 0.f;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int mt = 0; mt < K::MT; ++mt) {
        const int o = mt * 16 + hh * 8;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float gp = s.pz[2][o + r], gq = s.pz[3][o + r];
          const float zhp = (z[ct][mt][r] - mu2) * rs2, zhq = (z[ct][mt][r + 8] - mu2) * rs2;
          const float znp = fmaf(zhp, gp, s.pz[4][o + r]), znq = fmaf(zhq, gq, s.pz[5][o + r]);
          const float sg = rfx_sigmoid(znq);
          const float g = gr[ct][mt][r];
          ds_acc[mt][r] = fmaf(g, znp * sg, ds_acc[mt][r]);
          const float du = g * s.pz[6][o + r];
          const float dp = du * sg, dq = du * znp * sg * (1.0f - sg);
          g2w_acc[mt][r] = fmaf(dp, zhp, g2w_acc[mt][r]);
          g2w_acc[mt][r + 8] = fmaf(dq, zhq, g2w_acc[mt][r + 8]);
          g2b_acc[mt][r] += dp;
          g2b_acc[mt][r + 8] += dq;
          const float ep = gp * dp, eq = gq * dq;
          S1 += ep + eq;
          S2 = fmaf(ep, zhp, fmaf(eq, zhq, S2));
          z[ct][mt][r] = zhp; z[ct][mt][r + 8] = zhq;
          dzn[ct][mt][r] = dc_pack2(ep, eq);              // gamma * dzn: what the GroupNorm backward needs
        }
      }
    dc_wg_sum2(s.red, wave, lane, S1, S2, 0);
    const float m1 = S1 * inv2, m2 = S2 * inv2;
    // ---- phase B: dz -> global (bf16) and -> da = W2^T dz;  GELU', GroupNorm(1, H) backward sums
    f32x16 da[2];
    float Q1 = 0.f, Q2 = 0.f;
    uint16_t* dzp = a.dz + (int64_t)n * 2 * C * DC_T;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int t = 64 * wave + 32 * ct + j;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int mt = 0; mt < K::MT; ++mt) {
        float vp[8], vq[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const uint32_t pk = dzn[ct][mt][r];
          const float ep = __uint_as_float(pk << 16), eq = __uint_as_float(pk & 0xffff0000u);
          vp[r] = rs2 * (ep - m1 - z[ct][mt][r] * m2);
          vq[r] = rs2 * (eq - m1 - z[ct][mt][r + 8] * m2);
          const int c = dc_chan(mt, r, hh);
          dzp[(int64_t)c * DC_T + t] = (uint16_t)rfx_bf16_bits(vp[r]);
          dzp[(int64_t)(C + c) * DC_T + t] = (uint16_t)rfx_bf16_bits(vq[r]);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, s.w2tf[(mt * 2 + 0) * 64 + lane]), dc_frag8(vp), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, s.w2tf[(mt * 2 + 1) * 64 + lane]), dc_frag8(vq), acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = dc_hrow(r, hh);
        const float hh_ = (hacc[ct][r] - mu1) * rs1;                    // normalised h
        const float hn = fmaf(hh_, s.g1[m], s.be1[m]);
        const float dhn = m < K::H ? acc[r] * rfx_gelu_grad(hn) : 0.f;
        g1w_acc[r] = fmaf(dhn, hh_, g1w_acc[r]);
        g1b_acc[r] += dhn;
        const float e = s.g1[m] * dhn;
        Q1 += e; Q2 = fmaf(e, hh_, Q2);
        hacc[ct][r] = hh_;
        acc[r] = e;
      }
      da[ct] = acc;
    }
    dc_wg_sum2(s.red, wave, lane, Q1, Q2, 2);
    const float q1 = Q1 * inv1, q2 = Q2 * inv1;
    // ---- dh -> LDS (channels-last, for the taps) and -> global (bf16, for dW1)
    uint16_t* dhp = a.dh + (int64_t)n * K::H * DC_T;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int t = 64 * wave + 32 * ct + j;
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = dc_hrow(r, hh);
        v[r] = m < K::H ? rs1 * (da[ct][r] - q1 - hacc[ct][r] * q2) : 0.f;
        if (m < K::H) dhp[(int64_t)m * DC_T + t] = (uint16_t)rfx_bf16_bits(v[r]);
      }
      uint16_t* row = s.dhs + (t + DC_PADR) * L::DHP;
#pragma unroll
      for (int q = 0; q < 2 * K::NK2; ++q)       // run q = registers 4q..4q+3 = hidden rows 8q + 4hh .. + 3
        *reinterpret_cast<uint2*>(row + 8 * q + 4 * hh) = make_uint2(dc_pack2(v[4 * q], v[4 * q + 1]), dc_pack2(v[4 * q + 2], v[4 * q + 3]));
    }
    __syncthreads();
    // ---- dx = g + W1^T * dh, store; fetch the next sample's g
    float* op = a.out + (int64_t)n * C * DC_T;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int t = 64 * wave + 32 * ct + j;
#pragma unroll
      for (int mtx = 0; mtx < K::MTX; ++mtx) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 3; ++tap)
#pragma unroll
          for (int ks = 0; ks < K::NK2; ++ks) {
            const uint16_t* p = s.dhs + (t + DC_PADR - (tap - 1) * a.dil) * L::DHP + 16 * ks + 8 * hh;
            const bf16x8 b = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(p));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, s.w1tf[((mtx * 3 + tap) * K::NK2 + ks) * 64 + lane]), b, acc, 0, 0, 0);
          }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int mt = 2 * mtx + (r >> 3);
          if (mt < K::MT) op[(int64_t)dc_chan(mt, r & 7, hh) * DC_T + t] = gr[ct][mt][r & 7] + acc[r];
        }
      }
    }
    if (nn < a.N) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) dc_load_x<C>(a.g + (int64_t)nn * C * DC_T, 64 * wave + 32 * ct + j, hh, gr[ct]);
    }
  }
  // ---- reduce the per-lane accumulators: over the 32 lanes of a half-wave, then over waves; one row of `partial` per workgroup
  auto half_sum = [](float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  };
  __syncthreads();
  float* sc = &s.acc[wave][hh][0];
  int k = 0;
#pragma unroll
  for (int mt = 0; mt < K::MT; ++mt)
#pragma unroll
    for (int r = 0; r < 8; ++r) { const float v = half_sum(ds_acc[mt][r]); if (j == 0) sc[k] = v; ++k; }
#pragma unroll
  for (int mt = 0; mt < K::MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { const float v = half_sum(g2w_acc[mt][r]); if (j == 0) sc[k] = v; ++k; }
#pragma unroll
  for (int mt = 0; mt < K::MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { const float v = half_sum(g2b_acc[mt][r]); if (j == 0) sc[k] = v; ++k; }
#pragma unroll
  for (int r = 0; r < 16; ++r) { const float v = half_sum(g1w_acc[r]); if (j == 0) sc[k] = v; ++k; }
#pragma unroll
  for (int r = 0; r < 16; ++r) { const float v = half_sum(g1b_acc[r]); if (j == 0) sc[k] = v; ++k; }
  __syncthreads();
  // scratch index -> parameter index
  constexpr int NA = K::MT * 8 + 2 * K::MT * 16 + 32;
  float* out = a.partial + (int64_t)blockIdx.x * K::NPART;
  for (int idx = tid; idx < 2 * NA; idx += 256) {
    const int h2 = idx / NA, q = idx % NA;
    const float v = s.acc[0][h2][q] + s.acc[1][h2][q] + s.acc[2][h2][q] + s.acc[3][h2][q];
    if (q < K::MT * 8) {                                             // dscale[c]
      out[dc_chan(q >> 3, q & 7, h2)] = v;
    } else if (q < K::MT * 8 + 2 * K::MT * 16) {                     // dgn2w / dgn2b: value rows c, gate rows C + c
      const int u = q - K::MT * 8, which = u / (K::MT * 16), w = u % (K::MT * 16), mt = w >> 4, r = w & 15;
      const int c = dc_chan(mt, r & 7, h2);
      out[C + which * 2 * C + (r < 8 ? c : C + c)] = v;
    } else {                                                         // dgn1w / dgn1b by hidden row
      const int u = q - K::MT * 8 - 2 * K::MT * 16, which = u >> 4, m = dc_hrow(u & 15, h2);
      if (m < K::H) out[5 * C + which * K::H + m] = v;
    }
  }
}

template <int C>
static int dconv_launch_bwd(const DconvArgs& a, int grid, hipStream_t s) {
  const size_t lds = sizeof(DcLdsB<C>);
  if (lds > 160 * 1024) return -1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(dconv_bwd_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
      hipSuccess) return -3;
  hipLaunchKernelGGL(dconv_bwd_kernel<C>, dim3(grid), dim3(256), lds, s, a);
  RFX_CHECK_LAUNCH();
  return 0;
}

template <int C>
static int dconv_launch_fwd(const DconvArgs& a, hipStream_t s) {
  const size_t lds = sizeof(DcLds<C>);
  if (lds > 160 * 1024) return -1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(dconv_fwd_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
      hipSuccess) return -3;
  const int grid = a.N < 256 ? a.N : 256;
  hipLaunchKernelGGL(dconv_fwd_kernel<C>, dim3(grid), dim3(256), lds, s, a);
  RFX_CHECK_LAUNCH();
  return 0;
}

static bool dconv_ok(const DconvArgs& a, int C, int T) {
  return a.x && a.out && a.w1 && a.b1 && a.gn1w && a.gn1b && a.w2 && a.b2 && a.gn2w && a.gn2b && a.scale && a.N > 0 && T == DC_T &&
         (C == 48 || C == 96) && (a.dil == 1 || a.dil == 2);
}

extern "C" int rfx_dconv_layer_ok(int32_t C, int32_t T, int32_t dil) {
  return T == DC_T && (C == 48 || C == 96) && (dil == 1 || dil == 2);
}

extern "C" int rfx_dconv_layer_fwd(const float* x, float* out, int32_t N, int32_t C, int32_t T, int32_t dil, const float* w1,
                                   const float* b1, const float* gn1w, const float* gn1b, const float* w2, const float* b2,
                                   const float* gn2w, const float* gn2b, const float* scale, float eps, void* stream) {
  DconvArgs a{};
  a.x = x; a.out = out; a.w1 = w1; a.b1 = b1; a.gn1w = gn1w; a.gn1b = gn1b; a.w2 = w2; a.b2 = b2; a.gn2w = gn2w; a.gn2b = gn2b;
  a.scale = scale; a.N = N; a.dil = dil; a.eps = eps;
  if (!dconv_ok(a, C, T)) return -1;
  return C == 48 ? dconv_launch_fwd<48>(a, (hipStream_t)stream) : dconv_launch_fwd<96>(a, (hipStream_t)stream);
}

// partial: rfx_dconv_layer_bwd_rows(N) rows of 5C + 2H floats [dscale C | dgn2w 2C | dgn2b 2C | dgn1w H | dgn1b H]; the caller sums the rows.
extern "C" int rfx_dconv_layer_bwd_rows(int32_t N) { return N < 256 ? N : 256; }
extern "C" int rfx_dconv_layer_bwd(const float* x, const float* g, float* gx, int32_t N, int32_t C, int32_t T, int32_t dil,
                                   const float* w1, const float* b1, const float* gn1w, const float* gn1b, const float* w2,
                                   const float* b2, const float* gn2w, const float* gn2b, const float* scale, float eps,
                                   void* dz_bf16, float* a_out, void* dh_bf16, float* partial, void* stream) {
  DconvArgs a{};
  a.x = x; a.g = g; a.out = gx; a.w1 = w1; a.b1 = b1; a.gn1w = gn1w; a.gn1b = gn1b; a.w2 = w2; a.b2 = b2; a.gn2w = gn2w; a.gn2b = gn2b;
  a.scale = scale; a.N = N; a.dil = dil; a.eps = eps;
  a.dz = (uint16_t*)dz_bf16; a.a_out = a_out; a.dh = (uint16_t*)dh_bf16; a.partial = partial;
  if (!dconv_ok(a, C, T) || !g || !dz_bf16 || !a_out || !dh_bf16 || !partial || C != 48) return -1;
  return dconv_launch_bwd<48>(a, rfx_dconv_layer_bwd_rows(N), (hipStream_t)stream);
}
