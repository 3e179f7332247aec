// Fused DConv depth-layer on channels-last bf16 samples (rfx_cl_dconv_fwd / rfx_cl_dconv_bwd): the residual branch of the
// frequency encoder layers of Hybrid Demucs (torchaudio HDemucs `_DConv`, reached from remfx/models.py:319), bf16
// arithmetic mode, for samples x of [256 frames][C channels] (one (clip, frequency row) each):
//     h = conv1d(x; W1 (H, C, 3), b1, dilation d, padding d)            H = C / 4
//     a = GELU(GroupNorm(1, H)(h))
//     z = conv1d(a; W2 (2C, H, 1), b2)
//     y = x + scale[c] * GLU(GroupNorm(1, 2C)(z))
// One workgroup (8 waves) holds a sample; wave w owns positions [32 w, 32 w + 32).  The sample arrives by DMA as a dense
// [position][C] image -- the layout it has in memory.  All products run on v_mfma_f32_32x32x16_bf16:
//   * GEMM1  h[h][pos]  = W1 x        A = packed W1 fragments, B = the image (8 channels of a position = one 16-byte read per tap)
//   * GEMM2  z^T[pos][m] = a^T W2^T   A = GELU(GN(h)) straight from GEMM1's C/D registers (lane = position = A row; the packed W2
//     fragments enumerate k in the register order), B = packed W2^T.  The result has lane = CHANNEL, registers = positions: every
//     per-channel quantity (bias, affine, LayerScale, and in the backward pass the parameter-gradient sums) is per-lane, a GLU
//     pair (value tile t, gate tile t + NTV) sits in one lane, and the sample statistics are one cross-lane sum per sample.
//   * x (and gy, h in the backward pass) reach that same layout through an MFMA against identity fragments: exact for bf16.
// Backward (recompute z from the stored a; nothing of width 2C is read): GLU / LayerScale / GroupNorm-2 backward in registers,
// dz -> LDS image -> da^T = dz^T W2 (A = image rows), GELU / GroupNorm-1 backward, dh -> LDS image (with halo) ->
// dx^T = gy^T + sum_t dh^T(shifted) W1_t.  dz and dh also leave as channels-last tensors: the two weight-gradient GEMMs run on
// csrc/cl_wgrad.hip (deterministic).  The affine / LayerScale gradients are per-lane sums, reduced over waves and workgroups in a
// fixed order (no atomics: the LDS float atomics of the round-4 backward kernel cost 7 of its 13.7 ms).
#include "cl_common.h"
#include <stdlib.h>

#define CLD_T 256
// The packed-weight fragment reads are loop-invariant LDS loads: without a memory clobber in the sample loop LICM hoists all of them
// (52 - 64 registers of fragments, spilled to scratch on the spot and re-read from SCRATCH inside the loop: the first build of the
// backward kernel spent 80 % of its wave cycles parked on those reloads, SQ_WAIT_ANY / SQ_WAVE_CYCLES, r05 counters)
#define CLD_NO_HOIST() asm volatile("" ::: "memory")
#define CLD_HALO 2

struct ClDconvK {
  rfx_cl_dconv_desc d;
};

typedef short cld_s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ cl_bf16x8 cld_ld16(const unsigned char* p) {
  return __builtin_bit_cast(cl_bf16x8, *reinterpret_cast<const uint4*>(p));
}
// registers 8 s .. 8 s + 7 of a C/D tile as an MFMA A / B operand (k in register order)
__device__ __forceinline__ cl_bf16x8 cld_pack8(const float* v) {
  const uint4 u = make_uint4(rfx_cvt_pk_bf16(v[0], v[1]), rfx_cvt_pk_bf16(v[2], v[3]), rfx_cvt_pk_bf16(v[4], v[5]), rfx_cvt_pk_bf16(v[6], v[7]));
  return __builtin_bit_cast(cl_bf16x8, u);
}
// identity fragment u (k = 16 u + 8 khalf + e against column n): B operand that copies A's columns 16 u .. 16 u + 15 of a 32-column tile
__device__ __forceinline__ cl_bf16x8 cld_ident(int u, int lane) {
  const int n = lane & 31, k0 = 16 * u + 8 * (lane >> 5);
  uint32_t w[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) w[q] = (k0 + 2 * q == n ? 0x3f80u : 0u) | (k0 + 2 * q + 1 == n ? 0x3f800000u : 0u);
  return __builtin_bit_cast(cl_bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
}
__device__ __forceinline__ float cld_bf16r(float v) { return __uint_as_float(rfx_cvt_pk_bf16(v, 0.f) << 16); }
// sigmoid on the raw v_exp_f32 / v_rcp_f32 (no denormal fix-up sequence around the exponential: its argument is clamped instead)
__device__ __forceinline__ float cld_sigmoid(float v) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fminf(-1.44269504088896f * v, 126.0f)));
}

template <int C, int H>
struct CldCfg {
  static constexpr int HP = (H + 15) / 16 * 16, KC = C / 16, KH = HP / 16, NTV = (C + 31) / 32, NT2 = 2 * NTV, RH = 8 * KH;
  static constexpr int RS = 2 * C, RSH = 2 * HP, RSZ = 4 * C;                       // image row strides in bytes
  static constexpr int XIMG = (CLD_T + 2 * CLD_HALO) * RS;
  // forward LDS: x image | W1 A fragments (3 KC KiB) | W2^T B fragments (KH NT2 KiB) | reduction scratch
  static constexpr int F_W1 = XIMG, F_W2 = F_W1 + 3 * KC * 1024, F_RED = F_W2 + KH * NT2 * 1024, F_LDS = F_RED + 256;
  // backward LDS: gy image (later dx) | a image | h image | dz image | dh image (halo) | W2^T frags | W2 frags (da) | W1 frags (dx) | scratch
  static constexpr int B_A = CLD_T * RS, B_HI = B_A + CLD_T * RSH, B_DZ = B_HI + CLD_T * RSH, B_DH = B_DZ + CLD_T * RSZ,
                       B_W2 = B_DH + (CLD_T + 2 * CLD_HALO) * RSH, B_W2D = B_W2 + KH * NT2 * 1024, B_W1D = B_W2D + (2 * C / 16) * 1024,
                       B_RED = B_W1D + 3 * KH * NTV * 1024, B_LDS = B_RED + 8 * 64 * 4 * 0 + 512;
  static_assert(C % 16 == 0 && H * 4 == C && HP <= 32, "");
};

// cooperative copy of `bytes` (a multiple of 1024) from global to LDS, linear
__device__ __forceinline__ void cld_copy_in(unsigned char* lds, const void* src, int bytes, int tid, int nthreads = 512) {
  for (int o = tid * 16; o < bytes; o += nthreads * 16) *reinterpret_cast<uint4*>(lds + o) = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(src) + o);
}

// sum of v over the workgroup (all lanes get it); `red` = 8 floats of LDS; two barriers
template <int NW = 8>
__device__ __forceinline__ void cld_block_sum2(float& a, float& b, float* red, int wave, int lane) {
  a = rfx_wave_sum(a);
  b = rfx_wave_sum(b);
  if (lane == 0) { red[wave] = a; red[8 + wave] = b; }
  __syncthreads();
  float sa = 0.f, sb = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) { sa += red[w]; sb += red[8 + w]; }
  __syncthreads();
  a = sa; b = sb;
}

typedef float cld_f2 __attribute__((ext_vector_type(2)));
// erf-GELU on a register pair: cdf and exp(-x^2 / 2) as in rfx_gelu_parts (Abramowitz-Stegun 7.1.26), packed
__device__ __forceinline__ void cld_gelu_parts2(cld_f2 x, cld_f2& cdf, cld_f2& ex) {
  const cld_f2 ax = {fabsf(x[0]), fabsf(x[1])};
  const cld_f2 q = ax * (0.3275911f * 0.70710678118654752440f) + 1.0f;
  const cld_f2 t = {__builtin_amdgcn_rcpf(q[0]), __builtin_amdgcn_rcpf(q[1])};
  const cld_f2 xx = x * x * (-0.5f * 1.44269504088896f);
  ex = cld_f2{__builtin_amdgcn_exp2f(fmaxf(xx[0], -126.0f)), __builtin_amdgcn_exp2f(fmaxf(xx[1], -126.0f))};
  cld_f2 p = t * 1.061405429f + -1.453152027f;
  p = p * t + 1.421413741f;
  p = p * t + -0.284496736f;
  p = p * t + 0.254829592f;
  const cld_f2 e = 1.0f - (p * t) * ex;                  // erf(|x| / sqrt2)
  cdf = cld_f2{0.5f + copysignf(0.5f * e[0], x[0]), 0.5f + copysignf(0.5f * e[1], x[1])};
}
__device__ __forceinline__ cld_f2 cld_gelu2(cld_f2 x) { cld_f2 c, e; cld_gelu_parts2(x, c, e); return x * c; }
// GELU'(x) on a register pair
__device__ __forceinline__ cld_f2 cld_gelu_grad2(cld_f2 x) {
  cld_f2 cdf, ex;
  cld_gelu_parts2(x, cdf, ex);
  return (x * 0.39894228040143267794f) * ex + cdf;
}
// workgroup barrier that leaves this wave's VMEM operations (LDS-DMA pieces, output stores) in flight: __syncthreads() fences and
// drains them
#define CLD_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// LDS-DMA the compiler does not track: cl_glds16_quiet / cl_rsrc_words of cl_common.h
typedef cl_i32x4 cld_i32x4;
__device__ __forceinline__ cld_i32x4 cld_rsrc_words(const void* p, uint32_t bytes) { return cl_rsrc_words(p, bytes); }
__device__ __forceinline__ void cld_glds16_quiet(const cld_i32x4& rs, unsigned char* lds_base, uint32_t voff) { cl_glds16_quiet(rs, lds_base, voff); }
// 16 bytes of LDS, read and waited for inside one asm block
__device__ __forceinline__ uint4 cld_lds_read16(const unsigned char* p) {
  uint4 v;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p) : "memory");
  return v;
}
template <int N>
__device__ __forceinline__ void cld_wait_vm() {
  static_assert(N >= 0 && N <= 16, "");
  if (N == 0) CL_VMCNT(0); else if (N == 1) CL_VMCNT(1); else if (N == 2) CL_VMCNT(2); else if (N == 3) CL_VMCNT(3);
  else if (N == 4) CL_VMCNT(4); else if (N == 5) CL_VMCNT(5); else if (N == 6) CL_VMCNT(6); else if (N == 7) CL_VMCNT(7);
  else if (N == 8) CL_VMCNT(8); else if (N == 9) CL_VMCNT(9); else if (N == 10) CL_VMCNT(10); else if (N == 11) CL_VMCNT(11);
  else if (N == 12) CL_VMCNT(12); else if (N == 13) CL_VMCNT(13); else if (N == 14) CL_VMCNT(14); else if (N == 15) CL_VMCNT(15);
  else CL_VMCNT(16);
}

// the same with ONE raw barrier (s_barrier + lgkmcnt(0): VMEM operations stay in flight); `red` = 16 floats that no wave writes again
// before another workgroup barrier has passed
template <int NW = 8>
__device__ __forceinline__ void cld_block_sum2_raw(float& a, float& b, float* red, int wave, int lane) {
  a = rfx_wave_sum(a);
  b = rfx_wave_sum(b);
  if (lane == 0) { red[wave] = a; red[8 + wave] = b; }
  CLD_BARRIER();
  float sa = 0.f, sb = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) { sa += red[w]; sb += red[8 + w]; }
  a = sa; b = sb;
}

// PH: 0 = a sample is one 256-position tile, statistics inside the kernel (the frequency branch); 1 / 2 / 3 = a sample is TPS
// consecutive tiles (the time branch: a whole clip), GroupNorm statistics span all of them: pass 1 leaves the tile sums of h, pass 2
// (statistics 1 given) those of z, pass 3 (both given) finishes -- each pass recomputes the cheap front of the layer from x
// (K = 3 C and K = H GEMMs) instead of storing anything of width 2 C.  Tile edges inside a sample read the neighbour's rows.
template <int C, int H, int PH>
__global__ __launch_bounds__(512, 2) void cl_dconv_fwd_kernel(const ClDconvK g) {
  using Cfg = CldCfg<C, H>;
  constexpr int HP = Cfg::HP, KC = Cfg::KC, KH = Cfg::KH, NTV = Cfg::NTV, NT2 = Cfg::NT2, RH = Cfg::RH, RS = Cfg::RS;
  extern __shared__ __attribute__((aligned(16))) unsigned char cld_smem[];
  const rfx_cl_dconv_desc& d = g.d;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  // two x images: the next tile's rows arrive by untracked LDS-DMA while this one is worked on (section 4.13 of DESIGN.md: with the
  // builtin the compiler drains the DMA at the first LDS read it cannot prove disjoint); all in-loop barriers are raw, __syncthreads()
  // would drain it too
  constexpr int NVM = (PH == 0 || PH == 3) ? KC : 0;          // this wave's stores that are certain to follow a prefetch (the y rows)
  unsigned char* ximg = cld_smem;
  float* red = reinterpret_cast<float*>(cld_smem + Cfg::F_RED);
  const bool train = d.a != nullptr;

  // ---- once per workgroup: packed weights, zero halo rows
  cld_copy_in(cld_smem + Cfg::F_W1, d.w1p, 3 * KC * 1024, tid);
  cld_copy_in(cld_smem + Cfg::F_W2, d.w2p, KH * NT2 * 1024, tid);
  for (int o = tid * 4; o < CLD_HALO * RS; o += 512 * 4) {
    *reinterpret_cast<uint32_t*>(ximg + o) = 0u;
    *reinterpret_cast<uint32_t*>(ximg + (CLD_T + CLD_HALO) * RS + o) = 0u;
    *reinterpret_cast<uint32_t*>(cld_smem + Cfg::F_LDS + o) = 0u;
    *reinterpret_cast<uint32_t*>(cld_smem + Cfg::F_LDS + (CLD_T + CLD_HALO) * RS + o) = 0u;
  }
  // per-register parameters of the h tile (row h = (r & 3) + 8 (r >> 2) + 4 half), zero beyond H: padded rows stay exactly 0
  float b1r[RH], g1r[RH], e1r[RH];
#pragma unroll
  for (int r = 0; r < RH; ++r) {
    const int h = (r & 3) + 8 * (r >> 2) + 4 * half;
    b1r[r] = h < H ? d.b1[h] : 0.f;
    g1r[r] = h < H ? d.g1w[h] : 0.f;
    e1r[r] = h < H ? d.g1b[h] : 0.f;
  }
  // per-lane parameters of the z^T tiles: tile t < NTV = value channel 32 t + n, t >= NTV = its gate (W2 row C + channel)
  float b2v[NTV], b2g[NTV], gv[NTV], ev[NTV], gg[NTV], eg[NTV], sc[NTV];
  bool cok[NTV];
#pragma unroll
  for (int t = 0; t < NTV; ++t) {
    const int c = 32 * t + l31;
    cok[t] = c < C;
    b2v[t] = cok[t] ? d.b2[c] : 0.f;      b2g[t] = cok[t] ? d.b2[C + c] : 0.f;
    gv[t] = cok[t] ? d.g2w[c] : 0.f;      gg[t] = cok[t] ? d.g2w[C + c] : 0.f;
    ev[t] = cok[t] ? d.g2b[c] : 0.f;      eg[t] = cok[t] ? d.g2b[C + c] : 0.f;
    sc[t] = cok[t] ? d.scale[c] : 0.f;
  }
  const cl_bf16x8 id0 = cld_ident(0, lane), id1 = cld_ident(1, lane);
  const int p0 = 32 * wave;
  const unsigned char* xrow = ximg + (CLD_HALO + p0 + l31) * RS + 16 * half;          // this lane's position, channel half 8 * half
  const cld_i32x4 rs_xq = cld_rsrc_words(d.x, (uint32_t)min((int64_t)0x7ffffff0, (int64_t)d.S * CLD_T * RS));
  const float n1 = 1.0f / (H * CLD_T), n2 = 1.0f / (2 * C * CLD_T);
  // halo rows of a tile inside a multi-tile sample (PH != 0): waves 0 / 7 fetch the neighbouring tile's edge rows into registers one
  // tile ahead and put them down at the top of the tile they belong to; zeros at the sample's ends
  const bool halo_lane = PH != 0 && (wave == 0 || wave == 7) && lane < CLD_HALO * RS / 16;
  auto load_halo = [&](int s) {
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (halo_lane) {
      const int tile = s % d.TPS;
      const bool left = wave == 0;
      const bool inside = left ? tile > 0 : tile + 1 < d.TPS;
      if (inside)
        v = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(d.x) + (int64_t)s * (CLD_T * RS) +
                                            (left ? -(int64_t)CLD_HALO * RS : (int64_t)CLD_T * RS) + lane * 16);
    }
    return v;
  };
  // (PH >= 2) the sample's statistics, also one tile ahead: the wait the compiler puts in front of their use must not reach a prefetch
  auto load_st = [&](int s) { return *reinterpret_cast<const float4*>(d.stats + (int64_t)(s / d.TPS) * 4); };
  uint4 halo_nx = make_uint4(0u, 0u, 0u, 0u);
  float4 st_nx = make_float4(0.f, 0.f, 0.f, 0.f);
  if ((int)blockIdx.x < d.S) {
    const uint32_t sb0 = (uint32_t)blockIdx.x * (CLD_T * RS) + (uint32_t)p0 * RS + lane * 16;
    halo_nx = load_halo(blockIdx.x);
    if (PH >= 2) st_nx = load_st(blockIdx.x);
#pragma unroll
    for (int i = 0; i < KC; ++i) cld_glds16_quiet(rs_xq, ximg + (CLD_HALO + p0) * RS + i * 1024, sb0 + i * 1024);
    CL_VMCNT(0);
  }
  __syncthreads();

  int it = 0;
  float4 st_cur = st_nx;
  for (int s = blockIdx.x; s < d.S; s += gridDim.x, ++it) {
    CLD_NO_HOIST();
    {
      ximg = (it & 1) ? cld_smem + Cfg::F_LDS : cld_smem;
      xrow = ximg + (CLD_HALO + p0 + l31) * RS + 16 * half;
      if (halo_lane) *reinterpret_cast<uint4*>(ximg + (wave == 0 ? 0 : (CLD_T + CLD_HALO) * RS) + lane * 16) = halo_nx;
      // everything older than the previous sample's NVM output stores of this wave (its newest operations): this sample's DMA pieces
      cld_wait_vm<NVM>();
      CLD_BARRIER();                                            // the taps read the neighbouring waves' rows
      st_cur = st_nx;
      const int sn = s + (int)gridDim.x;
      if (sn < d.S) {
        unsigned char* xnext = (it & 1) ? cld_smem : cld_smem + Cfg::F_LDS;    // its rows: last read by this wave's own stores, two samples ago
        const uint32_t sbn = (uint32_t)sn * (CLD_T * RS) + (uint32_t)p0 * RS + lane * 16;
        halo_nx = load_halo(sn);
        if (PH >= 2) st_nx = load_st(sn);
#pragma unroll
        for (int i = 0; i < KC; ++i) cld_glds16_quiet(rs_xq, xnext + (CLD_HALO + p0) * RS + i * 1024, sbn + i * 1024);
      }
    }
    // ---- GEMM1
    f32x16 hacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) hacc[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int ks = 0; ks < KC; ++ks) {
        const cl_bf16x8 af = cld_ld16(cld_smem + Cfg::F_W1 + (t * KC + ks) * 1024 + lane * 16);
        const cl_bf16x8 bf = cld_ld16(xrow + (t - 1) * d.dil * RS + ks * 32);
        hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, hacc, 0, 0, 0);
      }
    float hv[RH];
    float s1, s2;
    {
      cld_f2 s1p = {0.f, 0.f}, s2p = {0.f, 0.f};
#pragma unroll
      for (int r = 0; r < RH; r += 2) {
        const cld_f2 h2 = cld_f2{hacc[r], hacc[r + 1]} + cld_f2{b1r[r], b1r[r + 1]};
        hv[r] = h2[0]; hv[r + 1] = h2[1];
        s1p += h2;
        s2p += h2 * h2;
      }
      s1 = s1p[0] + s1p[1]; s2 = s2p[0] + s2p[1];
    }
    float mu1, rs1;
    if (PH == 0) {
      cld_block_sum2_raw(s1, s2, red, wave, lane);
      mu1 = s1 * n1;
      rs1 = rsqrtf(fmaxf(s2 * n1 - mu1 * mu1, 0.f) + d.eps);
    } else if (PH == 1) {
      cld_block_sum2_raw(s1, s2, red + 16 * (it & 1), wave, lane);
      if (tid == 0) *reinterpret_cast<float2*>(d.partial + (int64_t)s * 2) = make_float2(s1, s2);
      continue;
    } else {
      mu1 = st_cur.x; rs1 = st_cur.y;
      CLD_BARRIER();                                 // the neighbours' tap reads of this wave's rows are done before they are overwritten
    }
    float av[RH];
    {
      const float k1 = -mu1 * rs1;
#pragma unroll
      for (int r = 0; r < RH; r += 2) {
        const cld_f2 hh = cld_f2{hv[r], hv[r + 1]} * rs1 + k1;
        const cld_f2 a2 = cld_gelu2(hh * cld_f2{g1r[r], g1r[r + 1]} + cld_f2{e1r[r], e1r[r + 1]});
        const uint32_t pk = rfx_cvt_pk_bf16(a2[0], a2[1]);       // the bf16 values GEMM2 multiplies (and the backward pass reads)
        av[r] = __uint_as_float(pk << 16); av[r + 1] = __uint_as_float(pk & 0xffff0000u);
      }
    }
    if (train && PH != 2) {
      // [pos][HP]: registers 4 q .. 4 q + 3 are rows 8 q + 4 half + 0..3 = 8 consecutive bytes
      uint16_t* ap = reinterpret_cast<uint16_t*>(d.a) + ((int64_t)s * CLD_T + p0 + l31) * HP + 4 * half;
      uint16_t* hp = reinterpret_cast<uint16_t*>(d.hpre) + ((int64_t)s * CLD_T + p0 + l31) * HP + 4 * half;
#pragma unroll
      for (int q = 0; q < RH / 4; ++q) {
        *reinterpret_cast<uint2*>(ap + 8 * q) = make_uint2(rfx_cvt_pk_bf16(av[4 * q], av[4 * q + 1]), rfx_cvt_pk_bf16(av[4 * q + 2], av[4 * q + 3]));
        *reinterpret_cast<uint2*>(hp + 8 * q) = make_uint2(rfx_cvt_pk_bf16(hv[4 * q], hv[4 * q + 1]), rfx_cvt_pk_bf16(hv[4 * q + 2], hv[4 * q + 3]));
      }
    }
    // ---- GEMM2 (transposed): z^T tiles, lane = channel
    f32x16 z[NT2];
#pragma unroll
    for (int t = 0; t < NT2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) z[t][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KH; ++ks) {
      const cl_bf16x8 af = cld_pack8(av + 8 * ks);
#pragma unroll
      for (int t = 0; t < NT2; ++t)
        z[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, cld_ld16(cld_smem + Cfg::F_W2 + (ks * NT2 + t) * 1024 + lane * 16), z[t], 0, 0, 0);
    }
    // (register pairs: v_pk_add / v_pk_fma_f32 do two elements per issue slot, and these passes are VALU-issue-bound.  Lanes of
    // channels >= C hold exact zeros: W2's padded columns and b2v = b2g = 0 there)
    {
      cld_f2 s1p = {0.f, 0.f}, s2p = {0.f, 0.f};
#pragma unroll
      for (int t = 0; t < NTV; ++t)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const cld_f2 zv = cld_f2{z[t][r], z[t][r + 1]} + b2v[t], zg = cld_f2{z[NTV + t][r], z[NTV + t][r + 1]} + b2g[t];
          z[t][r] = zv[0]; z[t][r + 1] = zv[1]; z[NTV + t][r] = zg[0]; z[NTV + t][r + 1] = zg[1];
          s1p += zv + zg;
          s2p += zv * zv + zg * zg;
        }
      s1 = s1p[0] + s1p[1]; s2 = s2p[0] + s2p[1];
    }
    float mu2, rs2;
    if (PH == 0) {
      cld_block_sum2_raw(s1, s2, red + 16, wave, lane);
      mu2 = s1 * n2;
      rs2 = rsqrtf(fmaxf(s2 * n2 - mu2 * mu2, 0.f) + d.eps);
      if (train && tid == 0) *reinterpret_cast<float4*>(d.stats + (int64_t)s * 4) = make_float4(mu1, rs1, mu2, rs2);
    } else if (PH == 2) {
      cld_block_sum2_raw(s1, s2, red + 16 * (it & 1), wave, lane);
      if (tid == 0) *reinterpret_cast<float2*>(d.partial + (int64_t)s * 2) = make_float2(s1, s2);
      continue;
    } else {
      mu2 = st_cur.z; rs2 = st_cur.w;
    }
    // ---- residual in the same layout, GLU, LayerScale; y over this wave's own rows of the image
#pragma unroll
    for (int t = 0; t < NTV; ++t) {
      f32x16 xr;
#pragma unroll
      for (int r = 0; r < 16; ++r) xr[r] = 0.f;
      xr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cld_ld16(xrow + (2 * t) * 32), id0, xr, 0, 0, 0);
      if (2 * t + 1 < KC) xr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cld_ld16(xrow + (2 * t + 1) * 32), id1, xr, 0, 0, 0);
      if (cok[t]) {
        // GroupNorm-2's normalisation and affine as one fused multiply-add per element: v = z (rstd g) + (e - mean rstd g); the
        // gate's pre-scaled by -log2(e) for the sigmoid's exp2
        const float av_ = rs2 * gv[t], bv_ = ev[t] - mu2 * av_;
        const float ag_ = -1.44269504088896f * rs2 * gg[t], bg_ = -1.44269504088896f * eg[t] - mu2 * ag_;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const cld_f2 v = cld_f2{z[t][r], z[t][r + 1]} * av_ + bv_, gn = cld_f2{z[NTV + t][r], z[NTV + t][r + 1]} * ag_ + bg_;
          const cld_f2 den = {1.0f + __builtin_amdgcn_exp2f(fminf(gn[0], 126.0f)), 1.0f + __builtin_amdgcn_exp2f(fminf(gn[1], 126.0f))};
          const cld_f2 sg = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
          const cld_f2 y = (v * sg) * sc[t] + cld_f2{xr[r], xr[r + 1]};
          xr[r] = y[0]; xr[r + 1] = y[1];
        }
      }
      // all lanes of the wave have read the x rows of tile t's channels (the MFMAs above) before they are overwritten
      __builtin_amdgcn_wave_barrier();
      if (cok[t]) {
        unsigned char* yb = ximg + (CLD_HALO + p0 + 4 * half) * RS + (32 * t + l31) * 2;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          *reinterpret_cast<uint16_t*>(yb + ((r & 3) + 8 * (r >> 2)) * RS) = (uint16_t)rfx_bf16_bits(xr[r]);
      }
    }
    CL_LGKM0();
    __builtin_amdgcn_wave_barrier();
    {
      unsigned char* yo = reinterpret_cast<unsigned char*>(d.y) + (int64_t)s * (CLD_T * RS) + p0 * RS + lane * 16;
      const unsigned char* yi = ximg + (CLD_HALO + p0) * RS + lane * 16;
#pragma unroll
      for (int i = 0; i < KC; ++i) *reinterpret_cast<uint4*>(yo + i * 1024) = *reinterpret_cast<const uint4*>(yi + i * 1024);
    }
    // the next sample's DMA overwrites this wave's rows: its own LDS reads above are done (program order + the waits), the
    // other waves' halo reads of them happened before the first statistics barrier of this sample
    CL_LGKM0();
  }
}

// Backward.  Per-lane parameter-gradient sums (lane = channel): LayerScale, GroupNorm-2 weight / bias (value and gate rows),
// GroupNorm-1 weight / bias; written per workgroup to `partial` ([gridDim.x][5 C + 2 H]: dscale | dgn2w | dgn2b | dgn1w | dgn1b)
// and added over workgroups in a fixed order by the second launch of rfx_cl_dconv_bwd.
// FOUR waves per workgroup, one per SIMD, each walking its 64 positions as two 32-position tiles: with eight waves of the SAME work
// split (256 registers each) the compiler spilled 104 registers and the waves sat parked on scratch reloads 80 % of their cycles
// (4.98 ms per launch at S = 32768, r05 SQ counters); one wave per SIMD has the whole 512-register file, what does not fit the 256
// VGPRs goes to AGPRs.  Round 6: kept as the reference form (RFX_DEV=1 RFX_CLD_BWD_NW=4) -- C = 48 runs cl_dconv_bwd8_kernel below,
// eight waves with the channel tiles split over wave pairs (2.49 -> 1.53 ms per launch).
template <int C, int H>
__global__ __launch_bounds__(256, 1) void cl_dconv_bwd_kernel(const ClDconvK g) {
  using Cfg = CldCfg<C, H>;
  constexpr int HP = Cfg::HP, KC = Cfg::KC, KH = Cfg::KH, NTV = Cfg::NTV, NT2 = Cfg::NT2, RS = Cfg::RS, RSH = Cfg::RSH, RSZ = Cfg::RSZ;
  constexpr int KZ = 2 * C / 16;                                  // K steps of da^T = dz^T W2 (k = the 2 C channels, natural order)
  constexpr int NW = 4, SUB = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char cld_smem[];
  const rfx_cl_dconv_desc& d = g.d;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  unsigned char* gimg = cld_smem;                                 // gy, later dx: [256][C]
  unsigned char* aimg = cld_smem + Cfg::B_A;                      // [256][HP]
  unsigned char* himg = cld_smem + Cfg::B_HI;                     // [256][HP]
  unsigned char* zimg = cld_smem + Cfg::B_DZ;                     // dz [256][2 C]
  unsigned char* dhimg = cld_smem + Cfg::B_DH;                    // dh [2 + 256 + 2][HP]
  float* red = reinterpret_cast<float*>(cld_smem + Cfg::B_RED);

  cld_copy_in(cld_smem + Cfg::B_W2, d.w2p, KH * NT2 * 1024, tid, 256);
  cld_copy_in(cld_smem + Cfg::B_W2D, d.w2dp, KZ * 1024, tid, 256);
  cld_copy_in(cld_smem + Cfg::B_W1D, d.w1dp, 3 * KH * NTV * 1024, tid, 256);
  for (int o = tid * 4; o < CLD_HALO * RSH; o += 256 * 4) {
    *reinterpret_cast<uint32_t*>(dhimg + o) = 0u;
    *reinterpret_cast<uint32_t*>(dhimg + (CLD_T + CLD_HALO) * RSH + o) = 0u;
  }
  float b2v[NTV], b2g[NTV], gv[NTV], ev[NTV], gg[NTV], eg[NTV], sc[NTV];
  bool cok[NTV];
#pragma unroll
  for (int t = 0; t < NTV; ++t) {
    const int c = 32 * t + l31;
    cok[t] = c < C;
    b2v[t] = cok[t] ? d.b2[c] : 0.f;      b2g[t] = cok[t] ? d.b2[C + c] : 0.f;
    gv[t] = cok[t] ? d.g2w[c] : 0.f;      gg[t] = cok[t] ? d.g2w[C + c] : 0.f;
    ev[t] = cok[t] ? d.g2b[c] : 0.f;      eg[t] = cok[t] ? d.g2b[C + c] : 0.f;
    sc[t] = cok[t] ? d.scale[c] : 0.f;
  }
  const bool hok = l31 < H;                                       // h-domain tiles: lane = hidden channel
  const float g1 = hok ? d.g1w[l31] : 0.f, e1 = hok ? d.g1b[l31] : 0.f;
  float a_ds[NTV], a_gwv[NTV], a_gwg[NTV], a_gbv[NTV], a_gbg[NTV], a_g1w = 0.f, a_g1b = 0.f;
#pragma unroll
  for (int t = 0; t < NTV; ++t) a_ds[t] = a_gwv[t] = a_gwg[t] = a_gbv[t] = a_gbg[t] = 0.f;
  const cl_bf16x8 id0 = cld_ident(0, lane), id1 = cld_ident(1, lane);
  const int w0 = 64 * wave;                                       // this wave's rows: [w0, w0 + 64)
  const int64_t big = 0x7ffffff0;
  const __amdgpu_buffer_rsrc_t rs_g = cl_rsrc(d.gy, (uint32_t)min(big, (int64_t)d.S * CLD_T * RS));
  const __amdgpu_buffer_rsrc_t rs_a = cl_rsrc(d.a, (uint32_t)min(big, (int64_t)d.S * CLD_T * RSH));
  const __amdgpu_buffer_rsrc_t rs_h = cl_rsrc(d.hpre, (uint32_t)min(big, (int64_t)d.S * CLD_T * RSH));
  const float n1 = 1.0f / (H * CLD_T), n2 = 1.0f / (2 * C * CLD_T);
  __syncthreads();

  for (int s = blockIdx.x; s < d.S; s += gridDim.x) {
    {
      const uint32_t gb = (uint32_t)s * (CLD_T * RS) + (uint32_t)w0 * RS + lane * 16;
#pragma unroll
      for (int i = 0; i < 2 * KC; ++i) cl_glds16(rs_g, gimg + w0 * RS + i * 1024, gb + i * 1024);
      const uint32_t hb = (uint32_t)s * (CLD_T * RSH) + (uint32_t)w0 * RSH + lane * 16;       // 64 rows x RSH bytes = 2 KH KiB
#pragma unroll
      for (int i = 0; i < 2 * KH; ++i) {
        cl_glds16(rs_a, aimg + w0 * RSH + i * 1024, hb + i * 1024);
        cl_glds16(rs_h, himg + w0 * RSH + i * 1024, hb + i * 1024);
      }
    }
    const float4 st = *reinterpret_cast<const float4*>(d.stats + (int64_t)s * 4);
    const float mu1 = st.x, rs1 = st.y, mu2 = st.z, rs2 = st.w;
    CL_VMCNT(0);
    __builtin_amdgcn_wave_barrier();                              // up to the halo barrier everything reads this wave's OWN rows
    // ---- pass A, one (value, gate) tile pair at a time: z^T recomputed from a (k in the register order of the forward pass: two
    // 8-byte runs per lane), GLU / LayerScale / GroupNorm-2 backward up to d(zhat), parked as bf16 in the dz image until the sample
    // sums are known
    float s1 = 0.f, s2 = 0.f;
    // (not unrolled: with both tiles' bodies in one block the parameter-gradient accumulators made the register allocator spill 182
    // registers past the 512 it has)
#pragma unroll 1
    for (int sub = 0; sub < SUB; ++sub) {
      const int p0 = w0 + 32 * sub, prow = p0 + 4 * half;
      const unsigned char* grow = gimg + (p0 + l31) * RS + 16 * half;
      cl_bf16x8 afr[KH];
#pragma unroll
      for (int ks = 0; ks < KH; ++ks) {
        const unsigned char* ar = aimg + (p0 + l31) * RSH + (16 * ks + 4 * half) * 2;
        const uint2 lo = *reinterpret_cast<const uint2*>(ar), hi = *reinterpret_cast<const uint2*>(ar + 16);
        afr[ks] = __builtin_bit_cast(cl_bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
      }
#pragma unroll
      for (int t = 0; t < NTV; ++t) {
        f32x16 zv, zg, gy;
#pragma unroll
        for (int r = 0; r < 16; ++r) zv[r] = zg[r] = gy[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KH; ++ks) {
          zv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks], cld_ld16(cld_smem + Cfg::B_W2 + (ks * NT2 + t) * 1024 + lane * 16), zv, 0, 0, 0);
          zg = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks], cld_ld16(cld_smem + Cfg::B_W2 + (ks * NT2 + NTV + t) * 1024 + lane * 16), zg, 0, 0, 0);
        }
        gy = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cld_ld16(grow + (2 * t) * 32), id0, gy, 0, 0, 0);
        if (2 * t + 1 < KC) gy = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cld_ld16(grow + (2 * t + 1) * 32), id1, gy, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float zhv = (zv[r] + b2v[t] - mu2) * rs2, zhg = (zg[r] + b2g[t] - mu2) * rs2;
          const float v = fmaf(zhv, gv[t], ev[t]), gt = fmaf(zhg, gg[t], eg[t]);
          const float sg = cld_sigmoid(gt);
          const float gyr = cok[t] ? gy[r] : 0.f;
          a_ds[t] = fmaf(gyr, v * sg, a_ds[t]);
          const float dg = gyr * sc[t];
          const float dv = dg * sg, dgt = dg * v * sg * (1.f - sg);
          a_gbv[t] += dv;  a_gwv[t] = fmaf(dv, zhv, a_gwv[t]);
          a_gbg[t] += dgt; a_gwg[t] = fmaf(dgt, zhg, a_gwg[t]);
          const uint32_t pk = rfx_cvt_pk_bf16(dv * gv[t], dgt * gg[t]);
          const float dzv = __uint_as_float(pk << 16), dzg = __uint_as_float(pk & 0xffff0000u);     // the parked values: the sums match them
          s1 += dzv + dzg;
          s2 = fmaf(dzv, zhv, fmaf(dzg, zhg, s2));
          if (cok[t]) {
            unsigned char* zb = zimg + (prow + (r & 3) + 8 * (r >> 2)) * RSZ + (32 * t + l31) * 2;
            *reinterpret_cast<uint16_t*>(zb) = (uint16_t)pk;
            *reinterpret_cast<uint16_t*>(zb + 2 * C) = (uint16_t)(pk >> 16);
          }
        }
      }
    }
    cld_block_sum2<NW>(s1, s2, red, wave, lane);
    // ---- pass B: zhat again (two MFMAs per tile), dz = rstd (d(zhat) - mean(d(zhat)) - zhat mean(d(zhat) zhat)) -> image; then
    // da^T = dz^T W2 (rows = own positions) and h^T through the identity: lane = hidden channel
    f32x16 da[SUB], ht[SUB];
    {
      const float m1 = s1 * n2, m2 = s2 * n2;
      s1 = 0.f; s2 = 0.f;
#pragma unroll
      for (int sub = 0; sub < SUB; ++sub) {
        const int p0 = w0 + 32 * sub, prow = p0 + 4 * half;
        cl_bf16x8 afr[KH];
#pragma unroll
        for (int ks = 0; ks < KH; ++ks) {
          const unsigned char* ar = aimg + (p0 + l31) * RSH + (16 * ks + 4 * half) * 2;
          const uint2 lo = *reinterpret_cast<const uint2*>(ar), hi = *reinterpret_cast<const uint2*>(ar + 16);
          afr[ks] = __builtin_bit_cast(cl_bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
        }
#pragma unroll
        for (int t = 0; t < NTV; ++t) {
          f32x16 zv, zg;
#pragma unroll
          for (int r = 0; r < 16; ++r) zv[r] = zg[r] = 0.f;
#pragma unroll
          for (int ks = 0; ks < KH; ++ks) {
            zv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks], cld_ld16(cld_smem + Cfg::B_W2 + (ks * NT2 + t) * 1024 + lane * 16), zv, 0, 0, 0);
            zg = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks], cld_ld16(cld_smem + Cfg::B_W2 + (ks * NT2 + NTV + t) * 1024 + lane * 16), zg, 0, 0, 0);
          }
          if (cok[t]) {
            unsigned char* zb = zimg + prow * RSZ + (32 * t + l31) * 2;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int ro = ((r & 3) + 8 * (r >> 2)) * RSZ;
              const float zhv = (zv[r] + b2v[t] - mu2) * rs2, zhg = (zg[r] + b2g[t] - mu2) * rs2;
              const float dzv = __uint_as_float((uint32_t)*reinterpret_cast<const uint16_t*>(zb + ro) << 16);
              const float dzg = __uint_as_float((uint32_t)*reinterpret_cast<const uint16_t*>(zb + ro + 2 * C) << 16);
              *reinterpret_cast<uint16_t*>(zb + ro) = (uint16_t)rfx_bf16_bits(rs2 * (dzv - m1 - zhv * m2));
              *reinterpret_cast<uint16_t*>(zb + ro + 2 * C) = (uint16_t)rfx_bf16_bits(rs2 * (dzg - m1 - zhg * m2));
            }
          }
        }
        f32x16 dat, htt;
#pragma unroll
        for (int r = 0; r < 16; ++r) dat[r] = htt[r] = 0.f;
#pragma unroll
        for (int kz = 0; kz < KZ; ++kz)
          dat = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cld_ld16(zimg + (p0 + l31) * RSZ + (16 * kz + 8 * half) * 2),
                                                         cld_ld16(cld_smem + Cfg::B_W2D + kz * 1024 + lane * 16), dat, 0, 0, 0);
        htt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cld_ld16(himg + (p0 + l31) * RSH + 16 * half), id0, htt, 0, 0, 0);
        if (KH > 1) htt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cld_ld16(himg + (p0 + l31) * RSH + 32 + 16 * half), id1, htt, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float hh = (htt[r] - mu1) * rs1;
          const float dhn = hok ? dat[r] * rfx_gelu_grad(fmaf(hh, g1, e1)) : 0.f;
          a_g1b += dhn;
          a_g1w = fmaf(dhn, hh, a_g1w);
          const float dhh = dhn * g1;
          s1 += dhh;
          s2 = fmaf(dhh, hh, s2);
          htt[r] = hh; dat[r] = dhh;
        }
        da[sub] = dat; ht[sub] = htt;
      }
    }
    cld_block_sum2<NW>(s1, s2, red, wave, lane);
    {
      const float m1 = s1 * n1, m2 = s2 * n1;
      if (l31 < HP) {
#pragma unroll
        for (int sub = 0; sub < SUB; ++sub) {
          unsigned char* hb = dhimg + (CLD_HALO + w0 + 32 * sub + 4 * half) * RSH + l31 * 2;
#pragma unroll
          for (int r = 0; r < 16; ++r)
            *reinterpret_cast<uint16_t*>(hb + ((r & 3) + 8 * (r >> 2)) * RSH) =
                hok ? (uint16_t)rfx_bf16_bits(rs1 * (da[sub][r] - m1 - ht[sub][r] * m2)) : (uint16_t)0;
        }
      }
    }
    __syncthreads();                                              // the taps read the neighbouring waves' rows of dh
    // ---- dx^T = gy^T + sum_t dh^T(pos - (t - 1) d) W1_t; written over gy (own rows)
#pragma unroll
    for (int sub = 0; sub < SUB; ++sub) {
      const int p0 = w0 + 32 * sub, prow = p0 + 4 * half;
      const unsigned char* grow = gimg + (p0 + l31) * RS + 16 * half;
#pragma unroll
      for (int t = 0; t < NTV; ++t) {
        f32x16 dx;
#pragma unroll
        for (int r = 0; r < 16; ++r) dx[r] = 0.f;
        dx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cld_ld16(grow + (2 * t) * 32), id0, dx, 0, 0, 0);
        if (2 * t + 1 < KC) dx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cld_ld16(grow + (2 * t + 1) * 32), id1, dx, 0, 0, 0);
#pragma unroll
        for (int tp = 0; tp < 3; ++tp)
#pragma unroll
          for (int ks = 0; ks < KH; ++ks)
            dx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cld_ld16(dhimg + (CLD_HALO + p0 + l31 - (tp - 1) * d.dil) * RSH + (16 * ks + 8 * half) * 2),
                                                          cld_ld16(cld_smem + Cfg::B_W1D + ((tp * KH + ks) * NTV + t) * 1024 + lane * 16), dx, 0, 0, 0);
        // every lane of the wave has read tile t's channels of these rows (the identity MFMAs) before they are overwritten: LDS
        // operations of one wave execute in order
        if (cok[t]) {
          unsigned char* xb = gimg + prow * RS + (32 * t + l31) * 2;
#pragma unroll
          for (int r = 0; r < 16; ++r) *reinterpret_cast<uint16_t*>(xb + ((r & 3) + 8 * (r >> 2)) * RS) = (uint16_t)rfx_bf16_bits(dx[r]);
        }
      }
    }
    CL_LGKM0();
    __builtin_amdgcn_wave_barrier();
    {
      unsigned char* o = reinterpret_cast<unsigned char*>(d.y) + (int64_t)s * (CLD_T * RS) + w0 * RS + lane * 16;
#pragma unroll
      for (int i = 0; i < 2 * KC; ++i) *reinterpret_cast<uint4*>(o + i * 1024) = *reinterpret_cast<const uint4*>(gimg + w0 * RS + i * 1024 + lane * 16);
      o = reinterpret_cast<unsigned char*>(d.dz) + (int64_t)s * (CLD_T * RSZ) + w0 * RSZ + lane * 16;
#pragma unroll
      for (int i = 0; i < 4 * KC; ++i) *reinterpret_cast<uint4*>(o + i * 1024) = *reinterpret_cast<const uint4*>(zimg + w0 * RSZ + i * 1024 + lane * 16);
      o = reinterpret_cast<unsigned char*>(d.dh) + (int64_t)s * (CLD_T * RSH) + w0 * RSH + lane * 16;
#pragma unroll
      for (int i = 0; i < 2 * KH; ++i) *reinterpret_cast<uint4*>(o + i * 1024) = *reinterpret_cast<const uint4*>(dhimg + (CLD_HALO + w0) * RSH + i * 1024 + lane * 16);
    }
    CL_LGKM0();
    // the next sample's dh rows are written only after its own block sums: every wave has finished its taps by then
  }

  // ---- parameter-gradient sums of this workgroup: lanes l and l + 32 hold the same channel, the waves different positions
  __syncthreads();
  float* acc = reinterpret_cast<float*>(zimg);                    // [NW waves][5 NTV + 2][64]
  constexpr int NQ = 5 * NTV + 2;
#pragma unroll
  for (int t = 0; t < NTV; ++t) {
    acc[(wave * NQ + 5 * t + 0) * 64 + lane] = a_ds[t];
    acc[(wave * NQ + 5 * t + 1) * 64 + lane] = a_gwv[t];
    acc[(wave * NQ + 5 * t + 2) * 64 + lane] = a_gwg[t];
    acc[(wave * NQ + 5 * t + 3) * 64 + lane] = a_gbv[t];
    acc[(wave * NQ + 5 * t + 4) * 64 + lane] = a_gbg[t];
  }
  acc[(wave * NQ + 5 * NTV) * 64 + lane] = a_g1w;
  acc[(wave * NQ + 5 * NTV + 1) * 64 + lane] = a_g1b;
  __syncthreads();
  float* prow_out = d.partial + (int64_t)blockIdx.x * (5 * C + 2 * H);
  for (int i = tid; i < 5 * C + 2 * H; i += 256) {
    // i -> (quantity q, lane n): dscale[c] | dgn2w[value c | gate c] | dgn2b[value c | gate c] | dgn1w[h] | dgn1b[h]
    int q, n;
    if (i < C) { q = 5 * (i >> 5) + 0; n = i & 31; }
    else if (i < 2 * C) { const int c = i - C; q = 5 * (c >> 5) + 1; n = c & 31; }
    else if (i < 3 * C) { const int c = i - 2 * C; q = 5 * (c >> 5) + 2; n = c & 31; }
    else if (i < 4 * C) { const int c = i - 3 * C; q = 5 * (c >> 5) + 3; n = c & 31; }
    else if (i < 5 * C) { const int c = i - 4 * C; q = 5 * (c >> 5) + 4; n = c & 31; }
    else if (i < 5 * C + H) { q = 5 * NTV; n = i - 5 * C; }
    else { q = 5 * NTV + 1; n = i - 5 * C - H; }
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) sum += acc[(w * NQ + q) * 64 + n] + acc[(w * NQ + q) * 64 + 32 + n];
    prow_out[i] = sum;
  }
}

// ---- the same backward pass on EIGHT waves (two per SIMD), C = 48 / 64 (two channel tiles) --------------------------------------
// cl_dconv_bwd_kernel above keeps 400 registers live per wave (four waves, one per SIMD): every LDS / MFMA / transcendental latency
// and the DMA wait at the top of a sample is exposed, and the kernel ran at 0.22 of HBM, VALU-issue-bound on paper but with ~45 %
// of the cycles idle.  Here a wave owns (64-position group pg, channel tile ts) in the channel-domain phases (pass A, the dz
// finalisation, dx) and the 32-position sub-tile 32 * wave in the hidden-channel phase (da, GELU / GroupNorm-1 backward): half the
// accumulators, constants and tiles per wave, < 256 registers, two waves per SIMD.  The price is workgroup barriers where the
// four-wave form had wave-private rows: after the DMA, between the dz finalisation and da (a row's 2 C channels come from both
// waves of a pair), and before the output stores.  Same arithmetic, same order of the block sums' terms within a wave.
template <int C, int H>
__global__ __launch_bounds__(512, 1) void cl_dconv_bwd8_kernel(const ClDconvK g) {
  using Cfg = CldCfg<C, H>;
  constexpr int HP = Cfg::HP, KC = Cfg::KC, KH = Cfg::KH, NTV = Cfg::NTV, NT2 = Cfg::NT2, RS = Cfg::RS, RSH = Cfg::RSH, RSZ = Cfg::RSZ;
  constexpr int KZ = 2 * C / 16;
  static_assert(NTV == 2, "one channel tile per wave of a pair");
  extern __shared__ __attribute__((aligned(16))) unsigned char cld_smem[];
  const rfx_cl_dconv_desc& d = g.d;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int pg = wave >> 1, ts = wave & 1;
  unsigned char* aimg = cld_smem + Cfg::B_A;                      // [256][HP]
  unsigned char* himg = cld_smem + Cfg::B_HI;                     // [256][HP]
  unsigned char* zimg = cld_smem + Cfg::B_DZ;                     // dz [256][2 C]
  unsigned char* dhimg = cld_smem + Cfg::B_DH;                    // dh [2 + 256 + 2][HP]
  float* red = reinterpret_cast<float*>(cld_smem + Cfg::B_RED);   // two block sums per sample, each with its own 16 floats

  cld_copy_in(cld_smem + Cfg::B_W2, d.w2p, KH * NT2 * 1024, tid, 512);
  cld_copy_in(cld_smem + Cfg::B_W2D, d.w2dp, KZ * 1024, tid, 512);
  cld_copy_in(cld_smem + Cfg::B_W1D, d.w1dp, 3 * KH * NTV * 1024, tid, 512);
  for (int o = tid * 4; o < CLD_HALO * RSH; o += 512 * 4) {
    *reinterpret_cast<uint32_t*>(dhimg + o) = 0u;
    *reinterpret_cast<uint32_t*>(dhimg + (CLD_T + CLD_HALO) * RSH + o) = 0u;
  }
  const int c = 32 * ts + l31;
  const bool cok = c < C;
  const float b2v = cok ? d.b2[c] : 0.f, b2g = cok ? d.b2[C + c] : 0.f;
  const float gv = cok ? d.g2w[c] : 0.f, gg = cok ? d.g2w[C + c] : 0.f;
  const float ev = cok ? d.g2b[c] : 0.f, eg = cok ? d.g2b[C + c] : 0.f;
  const float sc = cok ? d.scale[c] : 0.f;
  const bool hok = l31 < H;
  const float g1 = hok ? d.g1w[l31] : 0.f, e1 = hok ? d.g1b[l31] : 0.f;
  // the element-wise work runs on PAIRS of accumulator registers (rows r, r + 1 of a tile): v_pk_fma / v_pk_mul / v_pk_add_f32 do two
  // elements per issue slot and this kernel is VALU-issue-bound (60 VALU instructions per MFMA in the r05 counters of the scalar form)
  cld_f2 a_ds = {0.f, 0.f}, a_gwv = {0.f, 0.f}, a_gwg = {0.f, 0.f}, a_gbv = {0.f, 0.f}, a_gbg = {0.f, 0.f}, a_g1w = {0.f, 0.f}, a_g1b = {0.f, 0.f};
  const float ggn = -1.44269504088896f * gg, egn = -1.44269504088896f * eg;       // the gate's affine, pre-scaled for exp2(-x)
  const cl_bf16x8 id0 = cld_ident(0, lane), id1 = cld_ident(1, lane);
  // gy through the identity with the columns of channels >= C zeroed: those lanes then carry exact zeros through every sum
  const cl_bf16x8 zfrag = __builtin_bit_cast(cl_bf16x8, make_uint4(0u, 0u, 0u, 0u));
  const cl_bf16x8 idg0 = cok ? id0 : zfrag, idg1 = cok ? id1 : zfrag;
  const int r0 = 32 * wave;                                       // DMA / output rows and the hidden-phase sub-tile of this wave
  const int64_t big = 0x7ffffff0;
  const cld_i32x4 rs_g = cld_rsrc_words(d.gy, (uint32_t)min(big, (int64_t)d.S * CLD_T * RS));
  const cld_i32x4 rs_a = cld_rsrc_words(d.a, (uint32_t)min(big, (int64_t)d.S * CLD_T * RSH));
  const cld_i32x4 rs_h = cld_rsrc_words(d.hpre, (uint32_t)min(big, (int64_t)d.S * CLD_T * RSH));
  const float n1 = 1.0f / (H * CLD_T), n2 = 1.0f / (2 * C * CLD_T);
  // a sample's operands (this wave's 32 rows of gy, a, h) by DMA, its four statistics into registers
  auto fetch_g = [&](int s, unsigned char* gdst) {
    const uint32_t gb = (uint32_t)s * (CLD_T * RS) + (uint32_t)r0 * RS + lane * 16;           // 32 rows x RS bytes = KC KiB
#pragma unroll
    for (int i = 0; i < KC; ++i) cld_glds16_quiet(rs_g, gdst + r0 * RS + i * 1024, gb + i * 1024);
  };
  auto fetch_h = [&](int s, const cld_i32x4& rs, unsigned char* img) {
    const uint32_t hb = (uint32_t)s * (CLD_T * RSH) + (uint32_t)r0 * RSH + lane * 16;         // 32 rows x RSH bytes = KH KiB
#pragma unroll
    for (int i = 0; i < KH; ++i) cld_glds16_quiet(rs, img + r0 * RSH + i * 1024, hb + i * 1024);
  };
  constexpr int NSTORE = 3 * KC + KH;                             // 16-byte global stores per wave and sample
  float4 st_next = make_float4(0.f, 0.f, 0.f, 0.f);
  if ((int)blockIdx.x < d.S) {
    fetch_g(blockIdx.x, cld_smem);
    fetch_h(blockIdx.x, rs_a, aimg);
    fetch_h(blockIdx.x, rs_h, himg);
    st_next = *reinterpret_cast<const float4*>(d.stats + (int64_t)blockIdx.x * 4);
  }
  CL_VMCNT(0);
  __syncthreads();

  int it = 0;
  for (int s = blockIdx.x; s < d.S; s += gridDim.x, ++it) {
    CLD_NO_HOIST();
    // gy (later dx) alternates between two images: the next sample's operands arrive while this one's dx is formed and stored
    unsigned char* gimg = (it & 1) ? cld_smem + Cfg::B_LDS : cld_smem;
    unsigned char* gnext = (it & 1) ? cld_smem : cld_smem + Cfg::B_LDS;
    const float mu1 = st_next.x, rs1 = st_next.y, mu2 = st_next.z, rs2 = st_next.w;
    // everything older than the previous sample's output stores (the newest NSTORE operations of this wave): this sample's DMA pieces
    cld_wait_vm<NSTORE>();
    CLD_BARRIER();                                                // the pair's other wave fetched half of this wave's rows
    // the next sample's operands land while this one is worked on: gy now (the other image: free since the previous sample's stores),
    // a after the last pass that reads it, h after this wave's own read of its rows
    const int sn = s + (int)gridDim.x;
    const bool more = sn < d.S;
    if (more) fetch_g(sn, gnext);
    // ---- pass A: this wave's (value, gate) tile of z^T for the 64 positions of its group, GLU / LayerScale / GroupNorm-2 backward up to
    // d(zhat), parked as bf16 in the dz image until the sample sums are known
    const float kv = (b2v - mu2) * rs2, kg = (b2g - mu2) * rs2;   // zhat = z rstd + k
    cld_f2 s1p = {0.f, 0.f}, s2p = {0.f, 0.f};
#pragma unroll 1
    for (int sub = 0; sub < 2; ++sub) {
      const int p0 = 64 * pg + 32 * sub, prow = p0 + 4 * half;
      const unsigned char* grow = gimg + (p0 + l31) * RS + 16 * half;
      cl_bf16x8 afr[KH];
#pragma unroll
      for (int ks = 0; ks < KH; ++ks) {
        const unsigned char* ar = aimg + (p0 + l31) * RSH + (16 * ks + 4 * half) * 2;
        const uint2 lo = *reinterpret_cast<const uint2*>(ar), hi = *reinterpret_cast<const uint2*>(ar + 16);
        afr[ks] = __builtin_bit_cast(cl_bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
      }
      f32x16 zv, zg, gy;
#pragma unroll
      for (int r = 0; r < 16; ++r) zv[r] = zg[r] = gy[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KH; ++ks) {
        zv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks], cld_ld16(cld_smem + Cfg::B_W2 + (ks * NT2 + ts) * 1024 + lane * 16), zv, 0, 0, 0);
        zg = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks], cld_ld16(cld_smem + Cfg::B_W2 + (ks * NT2 + NTV + ts) * 1024 + lane * 16), zg, 0, 0, 0);
      }
      gy = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cld_ld16(grow + (2 * ts) * 32), idg0, gy, 0, 0, 0);
      if (2 * ts + 1 < KC) gy = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cld_ld16(grow + (2 * ts + 1) * 32), idg1, gy, 0, 0, 0);
      unsigned char* zb0 = zimg + prow * RSZ + c * 2;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const cld_f2 z_v = {zv[r], zv[r + 1]}, z_g = {zg[r], zg[r + 1]}, g2 = {gy[r], gy[r + 1]};
        const cld_f2 zhv = z_v * rs2 + kv, zhg = z_g * rs2 + kg;
        const cld_f2 v = zhv * gv + ev, gn = zhg * ggn + egn;
        const cld_f2 den = {1.0f + __builtin_amdgcn_exp2f(fminf(gn[0], 126.0f)), 1.0f + __builtin_amdgcn_exp2f(fminf(gn[1], 126.0f))};
        const cld_f2 sg = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
        a_ds += g2 * (v * sg);
        const cld_f2 dv = (g2 * sc) * sg, dgt = (dv * v) * (1.0f - sg);
        a_gbv += dv;  a_gwv += dv * zhv;
        a_gbg += dgt; a_gwg += dgt * zhg;
        const cld_f2 pv = dv * gv, pg = dgt * gg;
        const uint32_t pk0 = rfx_cvt_pk_bf16(pv[0], pg[0]), pk1 = rfx_cvt_pk_bf16(pv[1], pg[1]);
        // the parked values: the sums match them
        const cld_f2 dzv = {__uint_as_float(pk0 << 16), __uint_as_float(pk1 << 16)}, dzg = {__uint_as_float(pk0 & 0xffff0000u), __uint_as_float(pk1 & 0xffff0000u)};
        s1p += dzv + dzg;
        s2p += dzv * zhv + dzg * zhg;
        if (cok) {
          unsigned char* zb = zb0 + ((r & 3) + 8 * (r >> 2)) * RSZ;
          *reinterpret_cast<uint16_t*>(zb) = (uint16_t)pk0;
          *reinterpret_cast<uint16_t*>(zb + 2 * C) = (uint16_t)(pk0 >> 16);
          *reinterpret_cast<uint16_t*>(zb + RSZ) = (uint16_t)pk1;
          *reinterpret_cast<uint16_t*>(zb + RSZ + 2 * C) = (uint16_t)(pk1 >> 16);
        }
      }
    }
    float s1 = s1p[0] + s1p[1], s2 = s2p[0] + s2p[1];
    s1 = rfx_wave_sum(s1); s2 = rfx_wave_sum(s2);
    if (lane == 0) { red[wave] = s1; red[8 + wave] = s2; }
    CLD_BARRIER();
    {
      float sa = 0.f, sb = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) { sa += red[w]; sb += red[8 + w]; }
      s1 = sa; s2 = sb;
    }
    // ---- pass B1: zhat again, dz = rstd (d(zhat) - mean(d(zhat)) - zhat mean(d(zhat) zhat)) -> image (this wave's own elements)
    {
      const float m1 = s1 * n2, m2 = s2 * n2;
#pragma unroll 1
      for (int sub = 0; sub < 2; ++sub) {
        const int p0 = 64 * pg + 32 * sub, prow = p0 + 4 * half;
        cl_bf16x8 afr[KH];
#pragma unroll
        for (int ks = 0; ks < KH; ++ks) {
          const unsigned char* ar = aimg + (p0 + l31) * RSH + (16 * ks + 4 * half) * 2;
          const uint2 lo = *reinterpret_cast<const uint2*>(ar), hi = *reinterpret_cast<const uint2*>(ar + 16);
          afr[ks] = __builtin_bit_cast(cl_bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
        }
        f32x16 zv, zg;
#pragma unroll
        for (int r = 0; r < 16; ++r) zv[r] = zg[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KH; ++ks) {
          zv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks], cld_ld16(cld_smem + Cfg::B_W2 + (ks * NT2 + ts) * 1024 + lane * 16), zv, 0, 0, 0);
          zg = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks], cld_ld16(cld_smem + Cfg::B_W2 + (ks * NT2 + NTV + ts) * 1024 + lane * 16), zg, 0, 0, 0);
        }
        if (cok) {
          unsigned char* zb = zimg + prow * RSZ + c * 2;
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const int ro = ((r & 3) + 8 * (r >> 2)) * RSZ;
            const cld_f2 z_v = {zv[r], zv[r + 1]}, z_g = {zg[r], zg[r + 1]};
            const cld_f2 zhv = z_v * rs2 + kv, zhg = z_g * rs2 + kg;
            const cld_f2 dzv = {__uint_as_float((uint32_t)*reinterpret_cast<const uint16_t*>(zb + ro) << 16),
                                __uint_as_float((uint32_t)*reinterpret_cast<const uint16_t*>(zb + ro + RSZ) << 16)};
            const cld_f2 dzg = {__uint_as_float((uint32_t)*reinterpret_cast<const uint16_t*>(zb + ro + 2 * C) << 16),
                                __uint_as_float((uint32_t)*reinterpret_cast<const uint16_t*>(zb + ro + RSZ + 2 * C) << 16)};
            const cld_f2 ov = ((dzv - m1) - zhv * m2) * rs2, og = ((dzg - m1) - zhg * m2) * rs2;
            const uint32_t o0 = rfx_cvt_pk_bf16(ov[0], og[0]), o1 = rfx_cvt_pk_bf16(ov[1], og[1]);
            *reinterpret_cast<uint16_t*>(zb + ro) = (uint16_t)o0;
            *reinterpret_cast<uint16_t*>(zb + ro + 2 * C) = (uint16_t)(o0 >> 16);
            *reinterpret_cast<uint16_t*>(zb + ro + RSZ) = (uint16_t)o1;
            *reinterpret_cast<uint16_t*>(zb + ro + RSZ + 2 * C) = (uint16_t)(o1 >> 16);
          }
        }
      }
    }
    CLD_BARRIER();                                                // a row of dz = both tiles = both waves of the pair
    if (more) fetch_h(sn, rs_a, aimg);
    // ---- pass B2, sub-tile [r0, r0 + 32): da^T = dz^T W2 and h^T through the identity (lane = hidden channel), GELU / GroupNorm-1 backward
    f32x16 dat, htt;
    {
#pragma unroll
      for (int r = 0; r < 16; ++r) dat[r] = htt[r] = 0.f;
#pragma unroll
      for (int kz = 0; kz < KZ; ++kz)
        dat = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cld_ld16(zimg + (r0 + l31) * RSZ + (16 * kz + 8 * half) * 2),
                                                       cld_ld16(cld_smem + Cfg::B_W2D + kz * 1024 + lane * 16), dat, 0, 0, 0);
      htt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cld_ld16(himg + (r0 + l31) * RSH + 16 * half), id0, htt, 0, 0, 0);
      if (KH > 1) htt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cld_ld16(himg + (r0 + l31) * RSH + 32 + 16 * half), id1, htt, 0, 0, 0);
      const float k1 = -mu1 * rs1;
      s1p = cld_f2{0.f, 0.f}; s2p = cld_f2{0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const cld_f2 h2 = {htt[r], htt[r + 1]}, d2 = {dat[r], dat[r + 1]};
        const cld_f2 hh = h2 * rs1 + k1;
        const cld_f2 dhn = hok ? d2 * cld_gelu_grad2(hh * g1 + e1) : cld_f2{0.f, 0.f};
        a_g1b += dhn;
        a_g1w += dhn * hh;
        const cld_f2 dhh = dhn * g1;
        s1p += dhh;
        s2p += dhh * hh;
        htt[r] = hh[0]; htt[r + 1] = hh[1]; dat[r] = dhh[0]; dat[r + 1] = dhh[1];
      }
      s1 = s1p[0] + s1p[1]; s2 = s2p[0] + s2p[1];
    }
    if (more) {
      CL_LGKM0();                                                 // (this wave's reads of its h rows are long complete)
      fetch_h(sn, rs_h, himg);
      st_next = *reinterpret_cast<const float4*>(d.stats + (int64_t)sn * 4);
    }
    s1 = rfx_wave_sum(s1); s2 = rfx_wave_sum(s2);
    if (lane == 0) { red[16 + wave] = s1; red[24 + wave] = s2; }
    CLD_BARRIER();
    {
      float sa = 0.f, sb = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) { sa += red[16 + w]; sb += red[24 + w]; }
      const float m1 = sa * n1, m2 = sb * n1;
      if (l31 < HP) {
        unsigned char* hb = dhimg + (CLD_HALO + r0 + 4 * half) * RSH + l31 * 2;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          *reinterpret_cast<uint16_t*>(hb + ((r & 3) + 8 * (r >> 2)) * RSH) = hok ? (uint16_t)rfx_bf16_bits(rs1 * (dat[r] - m1 - htt[r] * m2)) : (uint16_t)0;
      }
    }
    CLD_BARRIER();                                                // the taps read the neighbouring waves' rows of dh
    // ---- dx^T = gy^T + sum_t dh^T(pos - (t - 1) d) W1_t; written over gy (this wave's tile of its group's rows)
#pragma unroll 1
    for (int sub = 0; sub < 2; ++sub) {
      const int p0 = 64 * pg + 32 * sub, prow = p0 + 4 * half;
      const unsigned char* grow = gimg + (p0 + l31) * RS + 16 * half;
      f32x16 dx;
#pragma unroll
      for (int r = 0; r < 16; ++r) dx[r] = 0.f;
      dx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cld_ld16(grow + (2 * ts) * 32), id0, dx, 0, 0, 0);
      if (2 * ts + 1 < KC) dx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cld_ld16(grow + (2 * ts + 1) * 32), id1, dx, 0, 0, 0);
#pragma unroll
      for (int tp = 0; tp < 3; ++tp)
#pragma unroll
        for (int ks = 0; ks < KH; ++ks)
          dx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cld_ld16(dhimg + (CLD_HALO + p0 + l31 - (tp - 1) * d.dil) * RSH + (16 * ks + 8 * half) * 2),
                                                        cld_ld16(cld_smem + Cfg::B_W1D + ((tp * KH + ks) * NTV + ts) * 1024 + lane * 16), dx, 0, 0, 0);
      // every lane of the wave has read this tile's channels of these rows (the identity MFMAs) before they are overwritten: LDS
      // operations of one wave execute in order, and the other wave of the pair reads and writes the other tile's channels only
      if (cok) {
        unsigned char* xb = gimg + prow * RS + c * 2;
#pragma unroll
        for (int r = 0; r < 16; ++r) *reinterpret_cast<uint16_t*>(xb + ((r & 3) + 8 * (r >> 2)) * RS) = (uint16_t)rfx_bf16_bits(dx[r]);
      }
    }
    CLD_BARRIER();                                                // rows [r0, r0 + 32) of dx: both waves of the pair
    {
      // (LDS reads the compiler cannot see: it would put `s_waitcnt vmcnt(0)` -- the next sample's DMA pieces -- in front of them)
      unsigned char* o = reinterpret_cast<unsigned char*>(d.y) + (int64_t)s * (CLD_T * RS) + r0 * RS + lane * 16;
#pragma unroll
      for (int i = 0; i < KC; ++i) *reinterpret_cast<uint4*>(o + i * 1024) = cld_lds_read16(gimg + r0 * RS + i * 1024 + lane * 16);
      o = reinterpret_cast<unsigned char*>(d.dz) + (int64_t)s * (CLD_T * RSZ) + r0 * RSZ + lane * 16;
#pragma unroll
      for (int i = 0; i < 2 * KC; ++i) *reinterpret_cast<uint4*>(o + i * 1024) = cld_lds_read16(zimg + r0 * RSZ + i * 1024 + lane * 16);
      o = reinterpret_cast<unsigned char*>(d.dh) + (int64_t)s * (CLD_T * RSH) + r0 * RSH + lane * 16;
#pragma unroll
      for (int i = 0; i < KH; ++i) *reinterpret_cast<uint4*>(o + i * 1024) = cld_lds_read16(dhimg + (CLD_HALO + r0) * RSH + i * 1024 + lane * 16);
    }
    CL_LGKM0();
    // the sample after the next lands in the rows of this gy image that this wave alone read last (its own output rows)
  }

  // ---- parameter-gradient sums of this workgroup: lanes l and l + 32 hold the same channel, the four waves of a tile different positions
  __syncthreads();
  float* acc = reinterpret_cast<float*>(zimg);                    // [8 waves][7][64]
  acc[(wave * 7 + 0) * 64 + lane] = a_ds[0] + a_ds[1];
  acc[(wave * 7 + 1) * 64 + lane] = a_gwv[0] + a_gwv[1];
  acc[(wave * 7 + 2) * 64 + lane] = a_gwg[0] + a_gwg[1];
  acc[(wave * 7 + 3) * 64 + lane] = a_gbv[0] + a_gbv[1];
  acc[(wave * 7 + 4) * 64 + lane] = a_gbg[0] + a_gbg[1];
  acc[(wave * 7 + 5) * 64 + lane] = a_g1w[0] + a_g1w[1];
  acc[(wave * 7 + 6) * 64 + lane] = a_g1b[0] + a_g1b[1];
  __syncthreads();
  float* prow_out = d.partial + (int64_t)blockIdx.x * (5 * C + 2 * H);
  for (int i = tid; i < 5 * C + 2 * H; i += 512) {
    // i -> (quantity q, channel tile t or -1, lane n): dscale[c] | dgn2w[value c | gate c] | dgn2b[value c | gate c] | dgn1w[h] | dgn1b[h]
    int q, t, n;
    if (i < 5 * C) { const int cc = i % C; q = i / C; t = cc >> 5; n = cc & 31; }
    else if (i < 5 * C + H) { q = 5; t = -1; n = i - 5 * C; }
    else { q = 6; t = -1; n = i - 5 * C - H; }
    float sum = 0.f;
    if (t >= 0) {
#pragma unroll
      for (int w = 0; w < 4; ++w) sum += acc[((2 * w + t) * 7 + q) * 64 + n] + acc[((2 * w + t) * 7 + q) * 64 + 32 + n];
    } else {
#pragma unroll
      for (int w = 0; w < 8; ++w) sum += acc[(w * 7 + q) * 64 + n] + acc[(w * 7 + q) * 64 + 32 + n];
    }
    prow_out[i] = sum;
  }
}

// per-wave LDS staging of the multi-pass backward kernel (cl_dconv_bwdp_kernel), by pass
template <int C, int H, int PASS>
struct CldPassLds {
  using Cfg = CldCfg<C, H>;
  static constexpr int RS = Cfg::RS, RSH = Cfg::RSH, RSZ = Cfg::RSZ;
  // pass 1: gy x 2 | a x 2 | z          pass 2: a | h | dh | z x 2
  static constexpr int O_G = 0, O_A = PASS == 1 ? 2 * 32 * RS : 0, O_H = O_A + 32 * RSH, O_DH = O_H + 32 * RSH;
  static constexpr int O_Z = PASS == 1 ? O_A + 2 * 32 * RSH : O_DH + 32 * RSH;
  static constexpr int WV = O_Z + (PASS == 1 ? 1 : 2) * 32 * RSZ;
  static constexpr int LDS = 4 * WV + Cfg::KH * Cfg::NT2 * 1024 + (PASS == 2 ? (2 * C / 16) * 1024 : 0) + 512;
};

// ---- backward in passes, for samples of several tiles (the time branch) and for widths whose single-pass images do not fit the LDS
// (C = 96).  Everything a pass needs from another tile is a per-sample scalar, reduced between the passes in a fixed order:
//   B1  z recomputed from a; GLU / LayerScale / GroupNorm-2 backward up to d(zhat); d(zhat) PARKED in the dz tensor (bf16); tile sums
//       of d(zhat) and d(zhat) zhat; per-lane sums for dscale / dgn2w / dgn2b
//   B2  (sample means given) zhat again, dz finalised in place; da^T = dz^T W2; GELU / GroupNorm-1 backward up to d(hhat), parked in
//       the dh tensor; tile sums; per-lane sums for dgn1w / dgn1b
//   B3  (means given) dh finalised in place (flat elementwise); dx = gy + conv^T(dh) then runs on cl_conv (taps across tile edges).
// Four waves, each staging its own 32-position sub-tiles in LDS (two per wave and tile).
template <int C, int H, int PASS>
__global__ __launch_bounds__(256, 1) void cl_dconv_bwdp_kernel(const ClDconvK g) {
  using Cfg = CldCfg<C, H>;
  constexpr int HP = Cfg::HP, KC = Cfg::KC, KH = Cfg::KH, NTV = Cfg::NTV, NT2 = Cfg::NT2, RS = Cfg::RS, RSH = Cfg::RSH, RSZ = Cfg::RSZ;
  constexpr int KZ = 2 * C / 16, NW = 4, SUB = 2;
  // LDS per wave -- pass 1: [gy sub-tile 32 RS] x 2 | [a sub-tile 32 RSH] x 2 | z sub-tile 32 RSZ (output)
  //                 pass 2: a | h | dh sub-tiles 32 RSH each | [z sub-tile 32 RSZ] x 2 (read, finalised in place, stored)
  // then the fragments.  The doubled buffers hold the NEXT sub-tile's operands: a wave is alone on its SIMD here (C = 96: the staging
  // of four waves fills the LDS; C = 48: 396 registers), so every DMA round trip it waits for is exposed -- 40 % of a 4 us sub-tile.
  // The pieces are issued by asm the compiler does not track and waited for with counted vmcnt.
  using L = CldPassLds<C, H, PASS>;
  constexpr int WV = L::WV, O_W2 = NW * WV, O_W2D = O_W2 + KH * NT2 * 1024, O_RED = O_W2D + (PASS == 2 ? KZ * 1024 : 0);
  extern __shared__ __attribute__((aligned(16))) unsigned char cld_smem[];
  const rfx_cl_dconv_desc& d = g.d;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  unsigned char* wv = cld_smem + wave * WV;
  unsigned char* const hsub = wv + L::O_H, *const dhsub = wv + L::O_DH;      // pass 2 only
  float* red = reinterpret_cast<float*>(cld_smem + O_RED);
  cld_copy_in(cld_smem + O_W2, d.w2p, KH * NT2 * 1024, tid, 256);
  if (PASS == 2) cld_copy_in(cld_smem + O_W2D, d.w2dp, KZ * 1024, tid, 256);
  float b2v[NTV], b2g[NTV], gv[NTV], ev[NTV], gg[NTV], eg[NTV], sc[NTV];
  bool cok[NTV];
#pragma unroll
  for (int t = 0; t < NTV; ++t) {
    const int c = 32 * t + l31;
    cok[t] = c < C;
    b2v[t] = cok[t] ? d.b2[c] : 0.f;      b2g[t] = cok[t] ? d.b2[C + c] : 0.f;
    gv[t] = cok[t] ? d.g2w[c] : 0.f;      gg[t] = cok[t] ? d.g2w[C + c] : 0.f;
    ev[t] = cok[t] ? d.g2b[c] : 0.f;      eg[t] = cok[t] ? d.g2b[C + c] : 0.f;
    sc[t] = cok[t] ? d.scale[c] : 0.f;
  }
  const bool hok = l31 < H;
  const float g1 = hok ? d.g1w[l31] : 0.f, e1 = hok ? d.g1b[l31] : 0.f;
  // element-wise work on register PAIRS (rows r, r + 1 of a tile): see cl_dconv_bwd8_kernel
  cld_f2 a_ds[NTV], a_gwv[NTV], a_gwg[NTV], a_gbv[NTV], a_gbg[NTV], a_g1w = {0.f, 0.f}, a_g1b = {0.f, 0.f};
  float ggn[NTV], egn[NTV];
#pragma unroll
  for (int t = 0; t < NTV; ++t) {
    a_ds[t] = a_gwv[t] = a_gwg[t] = a_gbv[t] = a_gbg[t] = cld_f2{0.f, 0.f};
    ggn[t] = -1.44269504088896f * gg[t]; egn[t] = -1.44269504088896f * eg[t];
  }
  const cl_bf16x8 id0 = cld_ident(0, lane), id1 = cld_ident(1, lane);
  const cl_bf16x8 zfrag = __builtin_bit_cast(cl_bf16x8, make_uint4(0u, 0u, 0u, 0u));
  const int64_t big = 0x7ffffff0;
  const cld_i32x4 rs_g = cld_rsrc_words(d.gy, (uint32_t)min(big, (int64_t)d.S * CLD_T * RS));
  const cld_i32x4 rs_a = cld_rsrc_words(d.a, (uint32_t)min(big, (int64_t)d.S * CLD_T * RSH));
  const cld_i32x4 rs_h = cld_rsrc_words(d.hpre, (uint32_t)min(big, (int64_t)d.S * CLD_T * RSH));
  const cld_i32x4 rs_z = cld_rsrc_words(d.dz, (uint32_t)min(big, (int64_t)d.S * CLD_T * RSZ));
  // operands of sub-tile `sub` of tile `s` (this wave's 32 positions), by kind; `par` = which of the doubled buffers
  auto fetch_a = [&](int s, int sub, int par) {
    const uint32_t hb = (uint32_t)s * (CLD_T * RSH) + (uint32_t)(64 * wave + 32 * sub) * RSH + lane * 16;
    unsigned char* dst = wv + L::O_A + (PASS == 1 ? par * 32 * RSH : 0);
#pragma unroll
    for (int i = 0; i < KH; ++i) cld_glds16_quiet(rs_a, dst + i * 1024, hb + i * 1024);
  };
  auto fetch_g = [&](int s, int sub, int par) {
    const uint32_t gb = (uint32_t)s * (CLD_T * RS) + (uint32_t)(64 * wave + 32 * sub) * RS + lane * 16;
#pragma unroll
    for (int i = 0; i < KC; ++i) cld_glds16_quiet(rs_g, wv + L::O_G + par * 32 * RS + i * 1024, gb + i * 1024);
  };
  auto fetch_h = [&](int s, int sub) {
    const uint32_t hb = (uint32_t)s * (CLD_T * RSH) + (uint32_t)(64 * wave + 32 * sub) * RSH + lane * 16;
#pragma unroll
    for (int i = 0; i < KH; ++i) cld_glds16_quiet(rs_h, hsub + i * 1024, hb + i * 1024);
  };
  auto fetch_z = [&](int s, int sub, int par) {
    const uint32_t zb = (uint32_t)s * (CLD_T * RSZ) + (uint32_t)(64 * wave + 32 * sub) * RSZ + lane * 16;
#pragma unroll
    for (int i = 0; i < 2 * KC; ++i) cld_glds16_quiet(rs_z, wv + L::O_Z + par * 32 * RSZ + i * 1024, zb + i * 1024);
  };
  // per-sample scalars travel one TILE ahead (loaded while the previous tile is worked on: the wait the compiler puts in front of their
  // first use then only reaches operations that are a tile old)
  auto load_st = [&](int s) { return *reinterpret_cast<const float4*>(d.stats + (int64_t)(s / d.TPS) * 4); };
  auto load_mm = [&](int s) { return *reinterpret_cast<const float2*>(d.sums + (int64_t)(s / d.TPS) * 4); };
  float4 st_nx = make_float4(0.f, 0.f, 0.f, 0.f);
  float2 mm_nx = make_float2(0.f, 0.f);
  if ((int)blockIdx.x < d.S) {
    st_nx = load_st(blockIdx.x);
    if (PASS == 2) mm_nx = load_mm(blockIdx.x);
    fetch_a(blockIdx.x, 0, 0);
    if (PASS == 1) fetch_g(blockIdx.x, 0, 0);
    else { fetch_h(blockIdx.x, 0); fetch_z(blockIdx.x, 0, 0); }
  }
  CL_VMCNT(0);
  __syncthreads();
  // global stores of one sub-tile, this wave: the newest operations in its queue when the next sub-tile starts
  constexpr int NSTORE = 2 * KC + (PASS == 2 ? KH : 0);
  unsigned parity = 0;                                             // the block sums of consecutive tiles alternate between two scratch rows

  for (int s = blockIdx.x; s < d.S; s += gridDim.x) {
    CLD_NO_HOIST();
    const float4 st = st_nx;
    const float mu1 = st.x, rs1 = st.y, mu2 = st.z, rs2 = st.w;
    const float m1 = mm_nx.x, m2 = mm_nx.y;
    const int s_nx = s + (int)gridDim.x;
    const bool more_tiles = s_nx < d.S;
    if (more_tiles) {
      st_nx = load_st(s_nx);
      if (PASS == 2) mm_nx = load_mm(s_nx);
    }
    cld_f2 s1p = {0.f, 0.f}, s2p = {0.f, 0.f};
    const float k1 = -mu1 * rs1;
#pragma unroll 1
    for (int sub = 0; sub < SUB; ++sub) {
      const int p0 = 64 * wave + 32 * sub;                       // first position of this sub-tile inside the tile
      const int prow = 4 * half;                                 // its rows inside the per-wave staging buffers
      // this sub-tile's operands: everything older than the previous sub-tile's stores (and whatever else came after them)
      cld_wait_vm<NSTORE>();
      __builtin_amdgcn_wave_barrier();
      const int nsub = sub ^ 1, ns = sub == 0 ? s : s_nx;        // the next sub-tile of this wave
      const bool more = sub == 0 || more_tiles;
      unsigned char* const asub = wv + L::O_A + (PASS == 1 ? sub * 32 * RSH : 0);
      unsigned char* const gsub = wv + L::O_G + sub * 32 * RS;                   // pass 1 only
      unsigned char* const zsub = wv + L::O_Z + (PASS == 2 ? sub * 32 * RSZ : 0);
      if (more) {
        if (PASS == 1) { fetch_a(ns, nsub, nsub); fetch_g(ns, nsub, nsub); }
        else fetch_z(ns, nsub, nsub);
      }
      cl_bf16x8 afr[KH];
#pragma unroll
      for (int ks = 0; ks < KH; ++ks) {
        const unsigned char* ar = asub + l31 * RSH + (16 * ks + 4 * half) * 2;
        const uint2 lo = *reinterpret_cast<const uint2*>(ar), hi = *reinterpret_cast<const uint2*>(ar + 16);
        afr[ks] = __builtin_bit_cast(cl_bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
      }
      if (PASS == 2 && more) {
        CL_LGKM0();                                                // the fragments of a are in registers: its buffer is free
        fetch_a(ns, nsub, 0);
      }
      const unsigned char* grow = gsub + l31 * RS + 16 * half;
#pragma unroll
      for (int t = 0; t < NTV; ++t) {
        f32x16 zv, zg;
#pragma unroll
        for (int r = 0; r < 16; ++r) zv[r] = zg[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KH; ++ks) {
          zv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks], cld_ld16(cld_smem + O_W2 + (ks * NT2 + t) * 1024 + lane * 16), zv, 0, 0, 0);
          zg = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks], cld_ld16(cld_smem + O_W2 + (ks * NT2 + NTV + t) * 1024 + lane * 16), zg, 0, 0, 0);
        }
        unsigned char* zb = zsub + prow * RSZ + (32 * t + l31) * 2;
        const float kv = (b2v[t] - mu2) * rs2, kg = (b2g[t] - mu2) * rs2;          // zhat = z rstd + k
        if (PASS == 1) {
          f32x16 gy;
#pragma unroll
          for (int r = 0; r < 16; ++r) gy[r] = 0.f;
          // (gy through the identity with the columns of channels >= C zeroed: exact zeros in those lanes through every sum)
          gy = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cld_ld16(grow + (2 * t) * 32), cok[t] ? id0 : zfrag, gy, 0, 0, 0);
          if (2 * t + 1 < KC) gy = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cld_ld16(grow + (2 * t + 1) * 32), cok[t] ? id1 : zfrag, gy, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const cld_f2 z_v = {zv[r], zv[r + 1]}, z_g = {zg[r], zg[r + 1]}, g2 = {gy[r], gy[r + 1]};
            const cld_f2 zhv = z_v * rs2 + kv, zhg = z_g * rs2 + kg;
            const cld_f2 v = zhv * gv[t] + ev[t], gn = zhg * ggn[t] + egn[t];
            const cld_f2 den = {1.0f + __builtin_amdgcn_exp2f(fminf(gn[0], 126.0f)), 1.0f + __builtin_amdgcn_exp2f(fminf(gn[1], 126.0f))};
            const cld_f2 sg = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
            a_ds[t] += g2 * (v * sg);
            const cld_f2 dv = (g2 * sc[t]) * sg, dgt = (dv * v) * (1.0f - sg);
            a_gbv[t] += dv;  a_gwv[t] += dv * zhv;
            a_gbg[t] += dgt; a_gwg[t] += dgt * zhg;
            const cld_f2 pv = dv * gv[t], pg = dgt * gg[t];
            const uint32_t pk0 = rfx_cvt_pk_bf16(pv[0], pg[0]), pk1 = rfx_cvt_pk_bf16(pv[1], pg[1]);
            const cld_f2 dzv = {__uint_as_float(pk0 << 16), __uint_as_float(pk1 << 16)}, dzg = {__uint_as_float(pk0 & 0xffff0000u), __uint_as_float(pk1 & 0xffff0000u)};
            s1p += dzv + dzg;
            s2p += dzv * zhv + dzg * zhg;
            if (cok[t]) {
              const int ro = ((r & 3) + 8 * (r >> 2)) * RSZ;
              *reinterpret_cast<uint16_t*>(zb + ro) = (uint16_t)pk0;
              *reinterpret_cast<uint16_t*>(zb + ro + 2 * C) = (uint16_t)(pk0 >> 16);
              *reinterpret_cast<uint16_t*>(zb + ro + RSZ) = (uint16_t)pk1;
              *reinterpret_cast<uint16_t*>(zb + ro + RSZ + 2 * C) = (uint16_t)(pk1 >> 16);
            }
          }
        } else if (cok[t]) {
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const int ro = ((r & 3) + 8 * (r >> 2)) * RSZ;
            const cld_f2 z_v = {zv[r], zv[r + 1]}, z_g = {zg[r], zg[r + 1]};
            const cld_f2 zhv = z_v * rs2 + kv, zhg = z_g * rs2 + kg;
            const cld_f2 dzv = {__uint_as_float((uint32_t)*reinterpret_cast<const uint16_t*>(zb + ro) << 16),
                                __uint_as_float((uint32_t)*reinterpret_cast<const uint16_t*>(zb + ro + RSZ) << 16)};
            const cld_f2 dzg = {__uint_as_float((uint32_t)*reinterpret_cast<const uint16_t*>(zb + ro + 2 * C) << 16),
                                __uint_as_float((uint32_t)*reinterpret_cast<const uint16_t*>(zb + ro + RSZ + 2 * C) << 16)};
            const cld_f2 ov = ((dzv - m1) - zhv * m2) * rs2, og = ((dzg - m1) - zhg * m2) * rs2;
            const uint32_t o0 = rfx_cvt_pk_bf16(ov[0], og[0]), o1 = rfx_cvt_pk_bf16(ov[1], og[1]);
            *reinterpret_cast<uint16_t*>(zb + ro) = (uint16_t)o0;
            *reinterpret_cast<uint16_t*>(zb + ro + 2 * C) = (uint16_t)(o0 >> 16);
            *reinterpret_cast<uint16_t*>(zb + ro + RSZ) = (uint16_t)o1;
            *reinterpret_cast<uint16_t*>(zb + ro + RSZ + 2 * C) = (uint16_t)(o1 >> 16);
          }
        }
      }
      if (PASS == 2) {
        f32x16 dat, htt;
#pragma unroll
        for (int r = 0; r < 16; ++r) dat[r] = htt[r] = 0.f;
#pragma unroll
        for (int kz = 0; kz < KZ; ++kz)
          dat = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cld_ld16(zsub + l31 * RSZ + (16 * kz + 8 * half) * 2),
                                                         cld_ld16(cld_smem + O_W2D + kz * 1024 + lane * 16), dat, 0, 0, 0);
        htt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cld_ld16(hsub + l31 * RSH + 16 * half), id0, htt, 0, 0, 0);
        if (KH > 1) htt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cld_ld16(hsub + l31 * RSH + 32 + 16 * half), id1, htt, 0, 0, 0);
        if (more) {
          CL_LGKM0();                                              // the MFMA operands above are in registers: the h buffer is free
          fetch_h(ns, nsub);
        }
        if (l31 < HP) {
          unsigned char* hb = dhsub + prow * RSH + l31 * 2;
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const cld_f2 h2 = {htt[r], htt[r + 1]}, d2 = {dat[r], dat[r + 1]};
            const cld_f2 hh = h2 * rs1 + k1;
            const cld_f2 dhn = hok ? d2 * cld_gelu_grad2(hh * g1 + e1) : cld_f2{0.f, 0.f};
            a_g1b += dhn;
            a_g1w += dhn * hh;
            const cld_f2 dp = dhn * g1;
            const uint32_t pk = rfx_cvt_pk_bf16(dp[0], dp[1]);                               // d(hhat) of rows r, r + 1, parked
            const cld_f2 dhh = {__uint_as_float(pk << 16), __uint_as_float(pk & 0xffff0000u)};
            s1p += dhh;
            s2p += dhh * hh;
            *reinterpret_cast<uint16_t*>(hb + ((r & 3) + 8 * (r >> 2)) * RSH) = (uint16_t)pk;
            *reinterpret_cast<uint16_t*>(hb + ((r & 3) + 8 * (r >> 2)) * RSH + RSH) = (uint16_t)(pk >> 16);
          }
        }
      }
      CL_LGKM0();
      __builtin_amdgcn_wave_barrier();
      {
        unsigned char* o = reinterpret_cast<unsigned char*>(d.dz) + (int64_t)s * (CLD_T * RSZ) + p0 * RSZ + lane * 16;
#pragma unroll
        for (int i = 0; i < 2 * KC; ++i) *reinterpret_cast<uint4*>(o + i * 1024) = *reinterpret_cast<const uint4*>(zsub + i * 1024 + lane * 16);
        if (PASS == 2) {
          o = reinterpret_cast<unsigned char*>(d.dh) + (int64_t)s * (CLD_T * RSH) + p0 * RSH + lane * 16;
#pragma unroll
          for (int i = 0; i < KH; ++i) *reinterpret_cast<uint4*>(o + i * 1024) = *reinterpret_cast<const uint4*>(dhsub + i * 1024 + lane * 16);
        }
      }
      CL_LGKM0();
    }
    float s1 = s1p[0] + s1p[1], s2 = s2p[0] + s2p[1];
    cld_block_sum2_raw<NW>(s1, s2, red + 16 * (int)(parity ^= 1), wave, lane);
    if (tid == 0) *reinterpret_cast<float2*>(d.tsum + (int64_t)s * 2) = make_float2(s1, s2);
  }

  // ---- parameter-gradient sums of this workgroup (pass 1: LayerScale / GroupNorm-2, pass 2: GroupNorm-1), as in the one-pass kernel
  __syncthreads();
  float* acc = reinterpret_cast<float*>(cld_smem);
  constexpr int NQ = 5 * NTV + 2;
#pragma unroll
  for (int t = 0; t < NTV; ++t) {
    acc[(wave * NQ + 5 * t + 0) * 64 + lane] = a_ds[t][0] + a_ds[t][1];
    acc[(wave * NQ + 5 * t + 1) * 64 + lane] = a_gwv[t][0] + a_gwv[t][1];
    acc[(wave * NQ + 5 * t + 2) * 64 + lane] = a_gwg[t][0] + a_gwg[t][1];
    acc[(wave * NQ + 5 * t + 3) * 64 + lane] = a_gbv[t][0] + a_gbv[t][1];
    acc[(wave * NQ + 5 * t + 4) * 64 + lane] = a_gbg[t][0] + a_gbg[t][1];
  }
  acc[(wave * NQ + 5 * NTV) * 64 + lane] = a_g1w[0] + a_g1w[1];
  acc[(wave * NQ + 5 * NTV + 1) * 64 + lane] = a_g1b[0] + a_g1b[1];
  __syncthreads();
  float* prow_out = d.partial + (int64_t)blockIdx.x * (5 * C + 2 * H);
  const int i_lo = PASS == 1 ? 0 : 5 * C, i_hi = PASS == 1 ? 5 * C : 5 * C + 2 * H;
  for (int i = i_lo + tid; i < i_hi; i += 256) {
    int q, n;
    if (i < C) { q = 5 * (i >> 5) + 0; n = i & 31; }
    else if (i < 2 * C) { const int c = i - C; q = 5 * (c >> 5) + 1; n = c & 31; }
    else if (i < 3 * C) { const int c = i - 2 * C; q = 5 * (c >> 5) + 2; n = c & 31; }
    else if (i < 4 * C) { const int c = i - 3 * C; q = 5 * (c >> 5) + 3; n = c & 31; }
    else if (i < 5 * C) { const int c = i - 4 * C; q = 5 * (c >> 5) + 4; n = c & 31; }
    else if (i < 5 * C + H) { q = 5 * NTV; n = i - 5 * C; }
    else { q = 5 * NTV + 1; n = i - 5 * C - H; }
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) sum += acc[(w * NQ + q) * 64 + n] + acc[(w * NQ + q) * 64 + 32 + n];
    prow_out[i] = sum;
  }
}

// per-sample means of the tile sums of a backward pass: sums[sample][2 which + {0, 1}] = (sum a, sum b) / n, fixed order
__global__ __launch_bounds__(256) void cl_dconv_means_kernel(const float* __restrict__ tsum, int nsamp, int TPS, double inv_n, int which,
                                                             float* __restrict__ sums) {
  const int smp = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
  if (smp >= nsamp) return;
  double a = 0.0, b = 0.0;
  for (int t = l; t < TPS; t += 64) {
    const float2 v = *reinterpret_cast<const float2*>(tsum + ((int64_t)smp * TPS + t) * 2);
    a += v.x; b += v.y;
  }
  a = rfx_wave_sum_d(a); b = rfx_wave_sum_d(b);
  if (l == 0) {
    sums[(int64_t)smp * 4 + 2 * which] = (float)(a * inv_n);
    sums[(int64_t)smp * 4 + 2 * which + 1] = (float)(b * inv_n);
  }
}

// pass B3: dh = rstd1 (d(hhat) - mean(d(hhat)) - hhat mean(d(hhat) hhat)) in place over the parked values; pad channels stay 0
__global__ __launch_bounds__(256) void cl_dconv_dh_kernel(uint16_t* __restrict__ dh, const uint16_t* __restrict__ hpre, const float* __restrict__ stats,
                                                          const float* __restrict__ sums, int64_t npos, int HP, int H, int64_t pos_per_sample) {
  const int g8 = HP / 8;
  const int64_t total = npos * g8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t p = i / g8;
    const int c0 = (int)(i - p * g8) * 8;
    const int64_t smp = p / pos_per_sample;
    const float mu1 = stats[smp * 4], rs1 = stats[smp * 4 + 1], m1 = sums[smp * 4 + 2], m2 = sums[smp * 4 + 3];
    float dv[8], hv[8], o[8];
    cl_unpack8(*reinterpret_cast<const uint4*>(dh + i * 8), dv);
    cl_unpack8(*reinterpret_cast<const uint4*>(hpre + i * 8), hv);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (c0 + e < H) ? rs1 * (dv[e] - m1 - (hv[e] - mu1) * rs1 * m2) : 0.f;
    *reinterpret_cast<uint4*>(dh + i * 8) = cl_pack8(o);
  }
}

// out[i] = sum over workgroups of partial[g][i], one wave per output, fixed order
struct CldPgDst { float* p[5]; };
// dst given: the sum is ADDED to the parameter's own gradient slice (segments dscale[C] | dgn2w[2C] | dgn2b[2C] | dgn1w[H] | dgn1b[H])
__global__ __launch_bounds__(256) void cl_dconv_pgrad_kernel(const float* __restrict__ partial, int n, int G, float* __restrict__ out,
                                                             const CldPgDst dst, int C, int H) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
  if (i >= n) return;
  float s = 0.f;
  for (int gidx = l; gidx < G; gidx += 64) s += partial[(int64_t)gidx * n + i];
  s = rfx_wave_sum(s);
  if (l != 0) return;
  if (dst.p[0] == nullptr) { out[i] = s; return; }
  if (i < C) dst.p[0][i] += s;
  else if (i < 3 * C) dst.p[1][i - C] += s;
  else if (i < 5 * C) dst.p[2][i - 3 * C] += s;
  else if (i < 5 * C + H) dst.p[3][i - 5 * C] += s;
  else dst.p[4][i - 5 * C - H] += s;
}

// statistics of multi-tile samples: one wave per sample adds the TPS tile sums in a fixed order (double), writes (mean, rstd) into
// fields (2 which, 2 which + 1) of stats[sample]
__global__ __launch_bounds__(256) void cl_dconv_stats_kernel(const float* __restrict__ part, int nsamp, int TPS, double inv_n, float eps, int which,
                                                             float* __restrict__ stats) {
  const int smp = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
  if (smp >= nsamp) return;
  double a = 0.0, b = 0.0;
  for (int t = l; t < TPS; t += 64) {
    const float2 v = *reinterpret_cast<const float2*>(part + ((int64_t)smp * TPS + t) * 2);
    a += v.x; b += v.y;
  }
  a = rfx_wave_sum_d(a); b = rfx_wave_sum_d(b);
  if (l == 0) {
    const double mu = a * inv_n, var = b * inv_n - mu * mu;
    stats[(int64_t)smp * 4 + 2 * which] = (float)mu;
    stats[(int64_t)smp * 4 + 2 * which + 1] = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)eps));
  }
}

template <int C, int H, int PH>
static int cld_launch_fwd(const rfx_cl_dconv_desc& d, hipStream_t st) {
  using Cfg = CldCfg<C, H>;
  constexpr int lds = Cfg::F_LDS + Cfg::XIMG;                      // a second x image for the prefetch
  static_assert(lds <= 160 * 1024, "");
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&cl_dconv_fwd_kernel<C, H, PH>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return -3;
    attr = true;
  }
  ClDconvK k;
  k.d = d;
  hipLaunchKernelGGL((cl_dconv_fwd_kernel<C, H, PH>), dim3(d.S < d.grid ? d.S : d.grid), dim3(512), lds, st, k);
  RFX_CHECK_LAUNCH();
  return 0;
}

template <int C, int H>
static int cld_forward(const rfx_cl_dconv_desc& d, hipStream_t st) {
  if (d.TPS <= 1) return cld_launch_fwd<C, H, 0>(d, st);
  const int nsamp = d.S / d.TPS;
  int rc = cld_launch_fwd<C, H, 1>(d, st);
  if (rc) return rc;
  hipLaunchKernelGGL(cl_dconv_stats_kernel, dim3((nsamp + 3) / 4), dim3(256), 0, st, d.partial, nsamp, d.TPS, 1.0 / ((double)H * CLD_T * d.TPS), d.eps, 0, d.stats);
  rc = cld_launch_fwd<C, H, 2>(d, st);
  if (rc) return rc;
  hipLaunchKernelGGL(cl_dconv_stats_kernel, dim3((nsamp + 3) / 4), dim3(256), 0, st, d.partial, nsamp, d.TPS, 1.0 / ((double)2 * C * CLD_T * d.TPS), d.eps, 1, d.stats);
  return cld_launch_fwd<C, H, 3>(d, st);
}

template <int C, int H>
static int cld_launch(const rfx_cl_dconv_desc& d, bool bwd, hipStream_t st) {
  using Cfg = CldCfg<C, H>;
  static bool attr = false;
  if (!bwd) return cld_forward<C, H>(d, st);
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&cl_dconv_bwd_kernel<C, H>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::B_LDS) != hipSuccess)
      return -3;
    attr = true;
  }
  ClDconvK k;
  k.d = d;
  if constexpr (Cfg::NTV == 2) {
    // dev A/B (honoured only under RFX_DEV=1): the four-wave form of round 5
    static const bool four = [] { const char* dv = getenv("RFX_DEV"); const char* e = getenv("RFX_CLD_BWD_NW"); return dv && atoi(dv) == 1 && e && atoi(e) == 4; }();
    static bool attr8 = false;
    constexpr int lds8 = Cfg::B_LDS + CLD_T * Cfg::RS;          // + the second gy image
    static_assert(lds8 <= 160 * 1024, "");
    if (!four) {
      if (!attr8) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&cl_dconv_bwd8_kernel<C, H>), hipFuncAttributeMaxDynamicSharedMemorySize, lds8) != hipSuccess)
          return -3;
        attr8 = true;
      }
      hipLaunchKernelGGL((cl_dconv_bwd8_kernel<C, H>), dim3(d.S < d.grid ? d.S : d.grid), dim3(512), lds8, st, k);
      RFX_CHECK_LAUNCH();
      return 0;
    }
  }
  hipLaunchKernelGGL((cl_dconv_bwd_kernel<C, H>), dim3(d.S < d.grid ? d.S : d.grid), dim3(256), Cfg::B_LDS, st, k);
  RFX_CHECK_LAUNCH();
  return 0;
}

static bool cld_common_ok(const rfx_cl_dconv_desc* d) {
  return d && d->x_or_gy_ok && d->S > 0 && (d->dil == 1 || d->dil == 2) && d->w2p && d->b1 && d->g1w && d->g1b && d->b2 && d->g2w && d->g2b &&
         d->scale && d->grid > 0 && (int64_t)d->S * CLD_T * 4 * d->C < 0x7ffffff0LL;
}

extern "C" int rfx_cl_dconv_ok(int32_t C, int32_t H, int32_t T, int32_t backward) {
  if (T <= 0 || T % CLD_T || H * 4 != C) return 0;          // T: positions per sample (a multiple of the 256-position tile)
  return C == 48 || C == 96;
}

extern "C" int rfx_cl_dconv_fwd(const rfx_cl_dconv_desc* dp, void* stream) {
  if (!dp || !dp->x || !dp->y || !dp->w1p) return -1;
  rfx_cl_dconv_desc d = *dp;
  d.x_or_gy_ok = 1;
  if (!cld_common_ok(&d) || !rfx_cl_dconv_ok(d.C, d.H, CLD_T, 0)) return -1;
  if (d.TPS < 1 || d.S % d.TPS) return -1;
  if ((d.a == nullptr) != (d.hpre == nullptr)) return -1;
  if (d.TPS == 1 ? (d.a == nullptr) != (d.stats == nullptr) : (!d.stats || !d.partial)) return -1;
  if (d.C == 48) return cld_launch<48, 12>(d, false, (hipStream_t)stream);
  return cld_launch<96, 24>(d, false, (hipStream_t)stream);
}

template <int C, int H, int PASS>
static int cld_launch_bwdp(const rfx_cl_dconv_desc& d, hipStream_t st) {
  using Cfg = CldCfg<C, H>;
  constexpr int lds = CldPassLds<C, H, PASS>::LDS;
  static_assert(lds <= 160 * 1024 && lds >= 4 * (5 * Cfg::NTV + 2) * 256, "");
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&cl_dconv_bwdp_kernel<C, H, PASS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return -3;
    attr = true;
  }
  ClDconvK k;
  k.d = d;
  hipLaunchKernelGGL((cl_dconv_bwdp_kernel<C, H, PASS>), dim3(d.S < d.grid ? d.S : d.grid), dim3(256), lds, st, k);
  RFX_CHECK_LAUNCH();
  return 0;
}

// backward in passes (see cl_dconv_bwdp_kernel); leaves dz and dh final, the parameter partials in d.partial; dx is the caller's
// (cl_conv: gy + transposed 3-tap convolution of dh)
template <int C, int H>
static int cld_backward_passes(const rfx_cl_dconv_desc& d, hipStream_t st) {
  const int nsamp = d.S / d.TPS;
  int rc = cld_launch_bwdp<C, H, 1>(d, st);
  if (rc) return rc;
  hipLaunchKernelGGL(cl_dconv_means_kernel, dim3((nsamp + 3) / 4), dim3(256), 0, st, d.tsum, nsamp, d.TPS, 1.0 / ((double)2 * C * CLD_T * d.TPS), 0, d.sums);
  rc = cld_launch_bwdp<C, H, 2>(d, st);
  if (rc) return rc;
  hipLaunchKernelGGL(cl_dconv_means_kernel, dim3((nsamp + 3) / 4), dim3(256), 0, st, d.tsum, nsamp, d.TPS, 1.0 / ((double)H * CLD_T * d.TPS), 1, d.sums);
  const int64_t npos = (int64_t)d.S * CLD_T;
  constexpr int HP = CldCfg<C, H>::HP;
  const int64_t tot = npos * (HP / 8);
  hipLaunchKernelGGL(cl_dconv_dh_kernel, dim3((unsigned)((tot + 255) / 256 < 8192 ? (tot + 255) / 256 : 8192)), dim3(256), 0, st,
                     reinterpret_cast<uint16_t*>(d.dh), reinterpret_cast<const uint16_t*>(d.hpre), d.stats, d.sums, npos, HP, H, (int64_t)d.TPS * CLD_T);
  RFX_CHECK_LAUNCH();
  return 0;
}

extern "C" int rfx_cl_dconv_bwd(const rfx_cl_dconv_desc* dp, float* pgrad, void* stream) {
  if (!dp || !dp->gy || !dp->a || !dp->hpre || !dp->stats || !dp->dz || !dp->dh || !dp->w2dp || !dp->partial) return -1;
  const bool direct = dp->pg_dst[0] != nullptr;
  for (int q = 0; q < 5; ++q)
    if ((dp->pg_dst[q] != nullptr) != direct) return -1;
  if (!direct && !pgrad) return -1;
  rfx_cl_dconv_desc d = *dp;
  d.x_or_gy_ok = 1;
  if (!cld_common_ok(&d) || !rfx_cl_dconv_ok(d.C, d.H, CLD_T, 1) || d.TPS < 1 || d.S % d.TPS) return -1;
  int rc;
  if (d.TPS > 1 || d.C != 48) {                       // passes: the caller computes dx (y unused)
    if (!d.tsum || !d.sums) return -1;
    rc = d.C == 48 ? cld_backward_passes<48, 12>(d, (hipStream_t)stream) : cld_backward_passes<96, 24>(d, (hipStream_t)stream);
  } else {
    if (!d.y || !d.w1dp) return -1;
    rc = cld_launch<48, 12>(d, true, (hipStream_t)stream);
  }
  if (rc) return rc;
  const int n = 5 * d.C + 2 * d.H, G = d.S < d.grid ? d.S : d.grid;
  CldPgDst dst;
  for (int q = 0; q < 5; ++q) dst.p[q] = d.pg_dst[q];
  hipLaunchKernelGGL(cl_dconv_pgrad_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, d.partial, n, G, pgrad, dst, d.C, d.H);
  RFX_CHECK_LAUNCH();
  return 0;
}
