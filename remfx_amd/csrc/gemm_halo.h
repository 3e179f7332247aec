// Halo-tile implicit GEMM for stride-1 multi-tap convolutions (bf16 arithmetic mode): the 3x3 rewrite convolutions of the
// Hybrid Demucs decoders and their input gradients, the 3-tap (dilated) 1-D convolutions of the time branch -- reached from
// remfx/models.py:319 (torchaudio HDemucs `_HDecLayer.rewrite`, `_DConv` conv).
//
// Why a second forward kernel.  gemm_tap_kernel gathers the B operand global -> VGPR once PER TAP: a 3x3 layer reads every input
// value nine times through the vector L1 (8 dword gathers + 4 converts per MFMA K step and lane), keeps two 32-position waves per
// SIMD fed through a 4-deep gather ring and synchronises the workgroup every 4 K steps for the A tile -- r03: 0.19 of the bf16
// MFMA peak with the machine to itself, latency-bound (SQ_WAIT_ANY 0.39).  Here a workgroup owns 128 consecutive output positions
// of one row and walks the reduction in 16-CHANNEL CHUNKS:
//   * the chunk's input HALO tile (rows x (128 + taps' column span) positions x 16 channels) is fetched ONCE, converted to bf16
//     and laid out channels-last in LDS (48-byte position stride: 16 channels + 16 bytes of padding = an odd multiple of 16 bytes,
//     conflict-free for ds_read_b128 in its 16-lane groups); every tap's B fragment is then ONE ds_read_b128 at a shifted
//     position -- no global access, no conversion, no bounds test inside the tap loop;
//   * the chunk's A tile (taps x 2 k8 rows x 32 R cells of packed weights, the same rfx_pack_a image the tap-major kernel reads:
//     K step (block, tap, sub-block) of the planner's channel-blocked order) sits next to it; both are written after ONE
//     barrier pair per chunk = per (taps x R) MFMAs and wave, against one per 4 R in the tap kernel;
//   * the global loads of chunk c + 1 (16 dwords per staged position + the A cells) are issued before the MFMA loop of chunk c
//     and consumed after it: their latency hides under 9 R MFMAs, three (R = 3) or four workgroups per CU cover each other's write
//     phases.
// The accumulator / tile context conventions are those of gemm_tap_kernel (wave w = positions [32 w, 32 w + 32) of the tile, all
// R channel tiles), so the epilogues of gemm_fwd.h (bias, activation, residual, GLU store, 16-bit pair stores, GroupNorm
// statistics) are reused unchanged.
#pragma once
#include "gemm_tap.h"

#define RFX_HALO_PS 48         // bytes per staged position: 16 bf16 channels + 16 bytes of padding (RFX_HALO_TW = 128 positions per
                               // workgroup and the launch-uniform eligibility test rfx_halo_geo_ok live in gemm_fwd.h)

template <int R, int NT, int IN16>
__global__ __launch_bounds__(256, R <= 2 ? 4 : 3) void gemm_halo_kernel(const FwdArgs g) {
  constexpr int BM = 32 * R, ACELLS = NT * 2 * BM, NSLOT = NT == 9 ? 2 : 1, NASLOT = (ACELLS + 255) / 256;
  constexpr int ESZ = IN16 ? 2 : 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char halo_smem[];
  uint4* a_lds = reinterpret_cast<uint4*>(halo_smem);                       // [NT][2][BM] cells
  unsigned char* img = halo_smem + ACELLS * 16;                             // [rows][W] positions x 48 bytes
  const rfx_gemm_desc& d = g.d;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int tpr = d.OB / RFX_HALO_TW, mtiles = d.Mpad / BM;
  const int64_t work = (int64_t)d.N * d.OA * tpr;
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int ym = q % mtiles;
  const int64_t pw = g.xcd_chunk > 0 ? (int64_t)xcd * g.xcd_chunk + q / mtiles : (int64_t)(q / mtiles) * 8 + xcd;
  if (pw >= work) return;
  const int n = (int)(pw / ((int64_t)d.OA * tpr));
  const int rem = (int)(pw - (int64_t)n * d.OA * tpr);
  const int a = rem / tpr, b0 = (rem - a * tpr) * RFX_HALO_TW;
  const int m0 = ym * BM;
  const int W = d.halo_w;

  // real taps -> byte offsets inside the image (wave-uniform: scalar loads of the first NT table rows)
  const int4* tab = reinterpret_cast<const int4*>(g.ktab);
  int tapoff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int4 e = tab[t];
    tapoff[t] = __builtin_amdgcn_readfirstlane(((e.y - d.halo_da0) * W + (e.z - d.halo_db0)) * RFX_HALO_PS);
  }
  // staging slots of this thread: one image position each (16 channel loads -> two 16-byte LDS stores)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(g.in) + (int64_t)n * d.in_ns * ESZ), 0, (int)d.in_extent, 0x00020000);
  uint32_t voff[NSLOT];
#pragma unroll
  for (int s = 0; s < NSLOT; ++s) {
    const int item = tid + 256 * s;
    const int r = item / W, c = item - r * W;
    const int ia = a + d.halo_da0 + r, ib = b0 + d.halo_db0 + c;
    const bool ok = (item < d.halo_rows * W) & ((unsigned)ia < (unsigned)d.IA) & ((unsigned)ib < (unsigned)d.IB);
    uint32_t off = (uint32_t)(((int64_t)ia * d.in_as + ib) * ESZ);
    asm volatile("" : "+v"(off));
    voff[s] = ok ? off : RFX_BUF_OOB;                // exactly 2^31: beyond num_records whatever the scalar offset adds
  }
  const uint32_t csb = (uint32_t)(d.in_cs * ESZ);    // channel stride in bytes
  // A staging slots: cell idx -> (tap t, k8 row kk8, column mm); K step of (chunk, tap t) = t * g2 + kbase(chunk).  The packed
  // matrix is one buffer: per-thread 32-bit byte offsets, the chunk's rows in the scalar offset
  const int g2 = d.gpt >> 1;
  const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.apack), 0, 0x7fffffff, 0x00020000);
  uint32_t aoff[NASLOT];
#pragma unroll
  for (int s = 0; s < NASLOT; ++s) {
    int idx = tid + 256 * s;
    idx = idx < ACELLS ? idx : ACELLS - 1;           // surplus threads re-read / re-write the last cell (same data)
    const int t = idx / (2 * BM), rem2 = idx - t * 2 * BM, kk8 = rem2 / BM, mm = rem2 - kk8 * BM;
    aoff[s] = (uint32_t)(((2 * t * g2 + kk8) * d.Mpad + m0 + mm) * 16);
  }
  const int nchunks = d.Kpad_t / (16 * NT);          // = channel blocks x sub-blocks of 16 channels

  f32x16 acc[R];
#pragma unroll
  for (int mt = 0; mt < R; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  float bpre[NSLOT][16];
  uint4 apre[NASLOT];
  auto issue = [&](int c) {                          // global loads of chunk c (clamped: the last iteration re-reads its own chunk)
    const int cc = __builtin_amdgcn_readfirstlane(c < nchunks ? c : nchunks - 1);
    const int blk = cc / g2, sub = cc - blk * g2;
    const uint32_t kb = (uint32_t)(2 * (blk * NT * g2 + sub) * d.Mpad * 16);
#pragma unroll
    for (int s = 0; s < NASLOT; ++s) apre[s] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsa, aoff[s], kb, 0));
    const uint32_t so = (uint32_t)(cc * 16) * csb;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (IN16)
          bpre[s][i] = __uint_as_float((uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rs, voff[s], so + (uint32_t)i * csb, 0));
        else
          bpre[s][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff[s], so + (uint32_t)i * csb, 0));
      }
  };
  issue(0);
  const unsigned char* bbase = img + (wave * 32 + l31) * RFX_HALO_PS + h * 16;
  const uint4* abase = a_lds + h * BM + l31;
  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();                                 // every wave is done reading the previous chunk's tiles
#pragma unroll
    for (int s = 0; s < NASLOT; ++s) {
      const int idx = tid + 256 * s;
      a_lds[(s + 1) * 256 <= ACELLS ? idx : (idx < ACELLS ? idx : ACELLS - 1)] = apre[s];
    }
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      const int item = tid + 256 * s;
      if (item < d.halo_rows * W) {
        float lo[8], hi[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { lo[i] = bpre[s][i]; hi[i] = bpre[s][8 + i]; }
        const bf16x8 f0 = IN16 ? bf16_pack8(lo) : round8(lo), f1 = IN16 ? bf16_pack8(hi) : round8(hi);
        *reinterpret_cast<uint4*>(img + item * RFX_HALO_PS) = __builtin_bit_cast(uint4, f0);
        *reinterpret_cast<uint4*>(img + item * RFX_HALO_PS + 16) = __builtin_bit_cast(uint4, f1);
      }
    }
    __syncthreads();
    issue(c + 1);
    __builtin_amdgcn_sched_barrier(0);               // the loads go out BEFORE the tap loop (hipcc otherwise sinks them behind two thirds of its MFMAs)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const bf16x8 bf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(bbase + tapoff[t]));
#pragma unroll
      for (int mt = 0; mt < R; ++mt) {
        const bf16x8 af = __builtin_bit_cast(bf16x8, abase[t * 2 * BM + mt * 32]);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[mt], 0, 0, 0);
      }
    }
  }
  TileCtx tc;
  tc.n = n; tc.pw = (int)(pw & 0x7fffffff); tc.m0 = m0; tc.wave = wave; tc.lane = lane; tc.l31 = l31; tc.h = h;
  tc.jvalid = true;
  tc.a = a; tc.b = b0 + wave * 32 + l31;
  fwd_epilogue_mid<R>(g, tc, acc);
  fwd_epilogue_store<R>(g, tc, acc);
}

template <int R, int NT, int IN16>
static int rfx_launch_halo_one(const FwdArgs& g, dim3 grid, hipStream_t s) {
  const size_t lds = (size_t)NT * 2 * 32 * R * 16 + (size_t)g.d.halo_rows * g.d.halo_w * RFX_HALO_PS;
  if (lds > 64 * 1024) return -1;
  hipLaunchKernelGGL((gemm_halo_kernel<R, NT, IN16>), grid, dim3(256), lds, s, g);
  RFX_CHECK_LAUNCH();
  return 0;
}

template <int IN16>
static int rfx_launch_halo_variant(const FwdArgs& g, int r, dim3 grid, hipStream_t s) {
  const bool nine = g.d.halo_nt == 9;
  switch (r) {
    case 1: return nine ? rfx_launch_halo_one<1, 9, IN16>(g, grid, s) : rfx_launch_halo_one<1, 3, IN16>(g, grid, s);
    case 2: return nine ? rfx_launch_halo_one<2, 9, IN16>(g, grid, s) : rfx_launch_halo_one<2, 3, IN16>(g, grid, s);
    case 3: return nine ? rfx_launch_halo_one<3, 9, IN16>(g, grid, s) : rfx_launch_halo_one<3, 3, IN16>(g, grid, s);
  }
  return -1;
}
