// Tap-major forward gather-GEMM on the bf16 matrix pipe of gfx950 (v_mfma_f32_32x32x16_bf16, fp32 accumulate):
// every convolution with >= 8 input channels in the "bf16x3" (fp32 operands split hi + lo, 3 MFMAs per product) and
// "bf16" (operands rounded to bf16, 1 MFMA: the bf16-mixed mode of BASELINE config 3) arithmetic modes.
//
// What changed against the channel-major kernel of gemm_fwd.h (which stays for exact fp32 and for Cin < 8):
//   * the reduction axis runs (tap, channel) in 8-CHANNEL GROUPS g = t * gpt + c8 (remfx_amd/convplan.py,
//     GemmPlan._build_tap_major).  K step ks = groups 2ks (lanes 0-31) and 2ks + 1 (lanes 32-63); a lane's MFMA B
//     fragment = 8 consecutive channels of ONE tap at its position.  So per K step a lane does one table read, ONE
//     bounds test and one offset add, then 8 raw buffer loads that differ by a multiple of the channel stride -- the
//     channel-major table cost a table row, two bounds tests and a select PER GATHER (11.5 VALU per MFMA in the r01
//     PMC run).  Out-of-range taps and the padded channels of the last group are handled by the hardware range check
//     of the buffer descriptor (num_records = one sample's extent): those loads return 0 without touching memory.
//   * the whole tap table (<= 112 taps + padding) sits in LDS for the kernel's lifetime; (tap, group) counters advance
//     per lane with two compare / subtract pairs.
//   * MODE 2 skips the hi / lo split: 4 v_cvt_pk_bf16_f32 per K step and one MFMA per channel tile, half the LDS for A.
// Unchanged: wave w owns 32 positions x all R channel tiles, gathers go global -> VGPR coalesced along the contiguous
// position axis (the B operand has no reuse across waves, so an LDS round trip would only add traffic), A (packed
// weights) is staged through LDS two K steps per barrier, gathers run 3 K steps ahead in a 4-deep register ring.
#pragma once
#include "gemm_fwd.h"

#define RFX_BDIST 3        // gather look-ahead in K steps
#define RFX_TAP_LDS 128    // tap-table slots in LDS (ntaps + 16 padding rows must fit)

struct TapLane {
  __amdgpu_buffer_rsrc_t rs;   // one sample of the operand; num_records = its extent in bytes
  uint32_t voff;               // byte offset of this lane's position inside the sample
  int ia0, ib0;                // input coordinates of the position (lanes beyond P: ia0 = 2^30, every bounds test fails)
  int t;                       // tap of this lane's NEXT gather ...
  uint32_t goff;               // ... and byte offset of its 8-channel group inside the tap (c8 * 8 * channel stride)
};

// x = hi + lo with hi = RNE_bf16(x), lo = RNE_bf16(x - hi): v_cvt_pk_bf16_f32 does two values per instruction and
// the residual is one packed subtract -> 5 VALU per pair
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
  uint32_t hw[4], lw[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x2_t v = {x[2 * q], x[2 * q + 1]};
    const uint32_t h = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
    const f32x2_t hf = {__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
    hw[q] = h;
    lw[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v - hf, bf16x2_t));
  }
  hi = __builtin_bit_cast(bf16x8, make_uint4(hw[0], hw[1], hw[2], hw[3]));
  lo = __builtin_bit_cast(bf16x8, make_uint4(lw[0], lw[1], lw[2], lw[3]));
}
__device__ __forceinline__ bf16x8 round8(const float (&x)[8]) {
  uint32_t w[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x2_t v = {x[2 * q], x[2 * q + 1]};
    w[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
  }
  return __builtin_bit_cast(bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
}

// eight zero-extended bf16 bit patterns (gather8_tap<true>) -> one B fragment
__device__ __forceinline__ bf16x8 bf16_pack8(const float (&x)[8]) {
  uint32_t w[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) w[q] = __float_as_uint(x[2 * q]) | (__float_as_uint(x[2 * q + 1]) << 16);
  return __builtin_bit_cast(bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
}

// The 8 gathers of one lane for one K step + advance of its (tap, group) position by two groups.
// gstep = 16 channel strides (two groups), gwrap = gpt * 8 channel strides (one tap), both in bytes.
// Branch-free on purpose: with a plain `ok ? offset : OOB` hipcc sank the offset arithmetic AND the table read into a
// branch (s_and_saveexec + ds_read + lgkmcnt(0) per K step); the empty asm makes the offset opaque, so it is computed
// unconditionally and the select stays a v_cndmask.  A masked lane must get EXACTLY 2^31: its raw offset may be
// "negative" (taps left of the row start), and 0xfffffff0 + i * cs4 would wrap back into the sample.
// IN16: 0 = fp32 operand, 1 = bf16 operand (8 two-byte loads), 3 = channels-last bf16 operand (probe).  (2 was a paired
// gather for 1x1 plans -- 4 dword loads + a DPP exchange: bit-exact, 0.9 ms slower on the step, removed)
template <int IN16 = 0>
__device__ __forceinline__ void gather8_tap(const rfx_gemm_desc& d, const int4* taps, uint32_t cs4, uint32_t gstep,
                                            uint32_t gwrap, TapLane& c, float (&b)[8]) {
  const int4 e = taps[c.t];          // (offset of channel 0, da, db, -): two distinct addresses per wave
  const bool ok = ((unsigned)(c.ia0 + e.y) < (unsigned)d.IA) & ((unsigned)(c.ib0 + e.z) < (unsigned)d.IB);
  uint32_t off = c.voff + ((uint32_t)e.x << (IN16 ? 1 : 2)) + c.goff;     // cs4 / gstep / gwrap / voff are BYTE quantities of the operand's type
  asm volatile("" : "+v"(off));
  // bit 31 set = beyond num_records (one sample spans < 2 GiB) = the load returns 0 and touches nothing
  const uint32_t base = ok ? off : RFX_BUF_OOB;
  if (IN16 == 3) {
    // channels-last bf16 operand (channel stride 1): the 8 channels of the group are 16 contiguous bytes -- ONE load per K step
    const uint4 v = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(c.rs, base, 0, 0));
    b[0] = __uint_as_float(v.x); b[1] = __uint_as_float(v.y); b[2] = __uint_as_float(v.z); b[3] = __uint_as_float(v.w);
  } else
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (IN16)      // bf16 operand: the 16 stored bits, zero-extended; bf16_pack8 pairs them up without any conversion
      b[i] = __uint_as_float((uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(c.rs, base + (uint32_t)i * cs4, 0, 0));
    else
      b[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c.rs, base + (uint32_t)i * cs4, 0, 0));
  }
  c.goff += gstep;
  bool w = c.goff >= gwrap;          // gpt >= 1: at most two wraps
  c.goff -= w ? gwrap : 0u; c.t += w ? 1 : 0;
  w = c.goff >= gwrap;
  c.goff -= w ? gwrap : 0u; c.t += w ? 1 : 0;
}

// A stage: NARR (hi [, lo]) x 2 k8 rows x BM cells of 16 bytes per K step, one or two per thread.
// Plain scalars (not arrays) so the two in-flight stages of the software pipeline stay in registers.
struct AStage { uint4 v0, v1; };
template <int R, int MODE>
__device__ __forceinline__ uint4 tap_a_cell(const uint4* __restrict__ apk, int64_t arr_stride, int Mpad, int k8_0,
                                            int m0, int idx) {
  constexpr int BM = 32 * R, NARR = MODE == 1 ? 2 : 1, NV = 2 * NARR * BM;
  idx = idx < NV ? idx : NV - 1;     // branch-free: surplus threads re-read the last cell
  const int arr = idx / (2 * BM), rem = idx % (2 * BM);
  const int kk8 = rem / BM, mm = rem % BM;
  return apk[arr * arr_stride + (int64_t)(k8_0 + kk8) * Mpad + m0 + mm];
}
template <int R, int MODE>
__device__ __forceinline__ AStage tap_a_load(const uint4* __restrict__ apk, int64_t arr_stride, int Mpad, int k8_0,
                                             int m0, int tid) {
  constexpr int NV = 2 * (MODE == 1 ? 2 : 1) * 32 * R;
  AStage s;
  s.v0 = tap_a_cell<R, MODE>(apk, arr_stride, Mpad, k8_0, m0, tid);
  s.v1 = NV > 256 ? tap_a_cell<R, MODE>(apk, arr_stride, Mpad, k8_0, m0, tid + 256) : s.v0;
  return s;
}
template <int R, int MODE>
__device__ __forceinline__ void tap_a_store(uint4* as, int tid, const AStage& s) {
  constexpr int NV = 2 * (MODE == 1 ? 2 : 1) * 32 * R;
  // UNCONDITIONAL stores (surplus threads rewrite the last cell with the same data): a store under a lane condition lets
  // LLVM sink the global load into that branch, right in front of a vmcnt(0)
  as[tid < NV ? tid : NV - 1] = s.v0;
  if (NV > 256) as[tid + 256 < NV ? tid + 256 : NV - 1] = s.v1;
}

// One 16-deep K step.  LDS holds the A tiles of FOUR K steps (a 64-deep K block) per buffer, so the workgroup barrier comes only
// after every fourth step (SUB == 3; rounds 2-3 had two steps per buffer: the K loop is latency-bound and every barrier pulls
// the four waves back into lockstep).  Per step ks:
//   LDS -> fragments of A(ks) from buffer (ks/4)&1, quarter SUB;
//   global -> registers: A tile of step ks+6 (into the register set that held A(ks+4)), gathers of step ks+3;
//   MFMAs;  registers -> LDS: A(ks+4) into the OTHER buffer, same quarter;  barrier if SUB == 3.
// Everything written in steps 4D .. 4D+3 is first read in step 4D+4, i.e. behind the barrier that ends step 4D+3, and overwrites
// what was last read in steps 4D-4 .. 4D-1, i.e. before the barrier that ended step 4D-1.
template <int R, int MODE, int SUB, int IN16 = 0>
__device__ __forceinline__ void k_step_tap(const rfx_gemm_desc& d, const uint4* __restrict__ apk, int64_t arr_stride,
                                           const int4* taps, uint32_t cs4, uint32_t gstep, uint32_t gwrap, int ks, int kmax, int m0, TapLane& c,
                                           uint4* as, f32x16 (&acc)[R], const float (&bc)[8], float (&bn)[8],
                                           AStage& a_set) {
  constexpr int BM = 32 * R, NARR = MODE == 1 ? 2 : 1, CELLS = 2 * NARR * BM;
  const int tid = threadIdx.x;
  const int lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int buf = (ks >> 2) & 1;
  const uint4* a_lds = as + buf * 4 * CELLS + SUB * CELLS;
  uint4 ah[R], al[R];
#pragma unroll
  for (int mt = 0; mt < R; ++mt) {
    ah[mt] = a_lds[h * BM + mt * 32 + l31];
    if (MODE == 1) al[mt] = a_lds[2 * BM + h * BM + mt * 32 + l31];
  }
  // issue order matters: vmcnt retires in order; the A tile is written to LDS two steps later, the gathers, consumed
  // RFX_BDIST steps later, go last and stay in flight
  const AStage a_now = a_set;                                         // A(ks+4), fetched two steps ago
  a_set = tap_a_load<R, MODE>(apk, arr_stride, d.Mpad, 2 * min(ks + 6, kmax), m0, tid);   // kmax: last K step of the padded pack
  gather8_tap<IN16>(d, taps, cs4, gstep, gwrap, c, bn);
  if (MODE == 1) {
    bf16x8 bh, bl;
    split8(bc, bh, bl);
#pragma unroll
    for (int mt = 0; mt < R; ++mt) {
      const bf16x8 fh = __builtin_bit_cast(bf16x8, ah[mt]), fl = __builtin_bit_cast(bf16x8, al[mt]);
      acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, bh, acc[mt], 0, 0, 0);
      acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, bl, acc[mt], 0, 0, 0);
      acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl, bh, acc[mt], 0, 0, 0);
    }
  } else {
    const bf16x8 bh = IN16 >= 2 ? __builtin_bit_cast(bf16x8, make_uint4(__float_as_uint(bc[0]), __float_as_uint(bc[1]), __float_as_uint(bc[2]),
                                                                       __float_as_uint(bc[3])))
                                : IN16 ? bf16_pack8(bc) : round8(bc);
#pragma unroll
    for (int mt = 0; mt < R; ++mt)
      acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[mt]), bh, acc[mt], 0, 0, 0);
  }
  tap_a_store<R, MODE>(as + (buf ^ 1) * 4 * CELLS + SUB * CELLS, tid, a_now);
  if (SUB == 3) __syncthreads();
  else { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }   // keep the two steps apart in the compiler too
}

template <int R, int MODE, int IN16 = 0>
__device__ __forceinline__ void run_phase_tap(const rfx_gemm_desc& d, const float* __restrict__ apack,
                                              const rfx_ktab_entry* __restrict__ tap_tab, int ntaps, int gpt, int Kpad, int m0,
                                              TapLane c, uint4* as, int4* taps, f32x16 (&acc)[R]) {
  constexpr int BM = 32 * R, NARR = MODE == 1 ? 2 : 1, CELLS = 2 * NARR * BM;
  const int tid = threadIdx.x;
  const int h = (tid & 63) >> 5;
  const int nk = Kpad / 16;
  if (nk == 0) return;
  const uint4* apk = reinterpret_cast<const uint4*>(apack);
  const int64_t arr_stride = (int64_t)(Kpad / 8 + 8) * d.Mpad;
  const uint32_t cs4 = (uint32_t)(d.in_cs * (IN16 ? 2 : 4));
  const uint32_t gstep = 16u * cs4, gwrap = (uint32_t)gpt * 8u * cs4;
  __syncthreads();            // a previous phase (two-phase launches) may still be reading the LDS buffers / tap table
  const int kmax = nk + 3;    // the pack carries 8 k8 rows (4 K steps) of padding behind Kpad
  {
    const AStage s0 = tap_a_load<R, MODE>(apk, arr_stride, d.Mpad, 0, m0, tid);
    const AStage s1 = tap_a_load<R, MODE>(apk, arr_stride, d.Mpad, 2, m0, tid);
    const AStage s2 = tap_a_load<R, MODE>(apk, arr_stride, d.Mpad, 4, m0, tid);
    const AStage s3 = tap_a_load<R, MODE>(apk, arr_stride, d.Mpad, 6, m0, tid);
    if (tid < ntaps + 16) taps[tid] = reinterpret_cast<const int4*>(tap_tab)[tid];   // table carries 16 invalid tail rows
    tap_a_store<R, MODE>(as, tid, s0);                       // A(0) .. A(3) -> buffer 0
    tap_a_store<R, MODE>(as + CELLS, tid, s1);
    tap_a_store<R, MODE>(as + 2 * CELLS, tid, s2);
    tap_a_store<R, MODE>(as + 3 * CELLS, tid, s3);
  }
  __syncthreads();
  c.t = h / gpt;              // group g = h of K step 0
  c.goff = (uint32_t)(h - c.t * gpt) * 8u * cs4;
  // the gathers run RFX_BDIST K steps ahead of the MFMAs: one K step is ~0.1-0.2 us of matrix work, a gather that misses
  // L2 takes ~1-2 us, and only two waves share a SIMD
  float b0[8], b1[8], b2[8], b3[8];
  AStage a0 = tap_a_load<R, MODE>(apk, arr_stride, d.Mpad, 2 * min(4, kmax), m0, tid);          // A(4), A(5): even / odd register set
  AStage a1 = tap_a_load<R, MODE>(apk, arr_stride, d.Mpad, 2 * min(5, kmax), m0, tid);
  gather8_tap<IN16>(d, taps, cs4, gstep, gwrap, c, b0);
  gather8_tap<IN16>(d, taps, cs4, gstep, gwrap, c, b1);
  gather8_tap<IN16>(d, taps, cs4, gstep, gwrap, c, b2);
  int ks = 0;
  for (; ks + 3 < nk; ks += 4) {
    k_step_tap<R, MODE, 0, IN16>(d, apk, arr_stride, taps, cs4, gstep, gwrap, ks, kmax, m0, c, as, acc, b0, b3, a0);
    k_step_tap<R, MODE, 1, IN16>(d, apk, arr_stride, taps, cs4, gstep, gwrap, ks + 1, kmax, m0, c, as, acc, b1, b0, a1);
    k_step_tap<R, MODE, 2, IN16>(d, apk, arr_stride, taps, cs4, gstep, gwrap, ks + 2, kmax, m0, c, as, acc, b2, b1, a0);
    k_step_tap<R, MODE, 3, IN16>(d, apk, arr_stride, taps, cs4, gstep, gwrap, ks + 3, kmax, m0, c, as, acc, b3, b2, a1);
  }
  if (ks < nk) k_step_tap<R, MODE, 0, IN16>(d, apk, arr_stride, taps, cs4, gstep, gwrap, ks, kmax, m0, c, as, acc, b0, b3, a0);
  if (ks + 1 < nk) k_step_tap<R, MODE, 1, IN16>(d, apk, arr_stride, taps, cs4, gstep, gwrap, ks + 1, kmax, m0, c, as, acc, b1, b0, a1);
  if (ks + 2 < nk) k_step_tap<R, MODE, 2, IN16>(d, apk, arr_stride, taps, cs4, gstep, gwrap, ks + 2, kmax, m0, c, as, acc, b2, b1, a0);
}

// Occupancy of the bf16 mode: four waves per SIMD for R <= 3 (128 VGPRs), three for R = 4 (168 VGPRs).  For R <= 2 and R = 4 the K
// loop fits those budgets without scratch (checked in the ISA: accumulators 16 R + gather ring 32 + A fragments 4 R + staging); what
// spills is the general epilogue, once per tile -- and the full tiles take the lean store anyway.  More resident waves hide more
// gather latency: same-box A/B of the Demucs step (r03): R = 3 at three waves -1.5 ms, R = 4 at three waves -1.2 ms, R <= 2 at four
// waves -0.7 ms, R = 3 at four waves another -1.0 ms (540 B of scratch per lane, still a net gain).  One step further the spills land
// in the K loop and the step collapses: R = 4 at four waves 147 -> 379 ms, R = 3 at five 336 ms, R <= 2 at five 156 ms (second builds
// of the library, same box).  (The split-bf16x3 mode holds hi + lo fragments and stays at two waves: R <= 2 at three is neutral,
// 206.3 vs 206.1 ms on the bf16x3 step, R = 3, 4 at three +7.7 ms.)
template <int R, int MODE, int IN16 = 0>
__global__ __launch_bounds__(256, MODE == 2 ? (R <= 3 ? 4 : 3) : 2) void gemm_tap_kernel(const FwdArgs g) {
  constexpr int BM = 32 * R, NARR = MODE == 1 ? 2 : 1, CELLS = 2 * NARR * BM;
  __shared__ __attribute__((aligned(16))) uint4 smem[8 * CELLS + RFX_TAP_LDS];   // A: 2 buffers x 4 K steps; tap table
  uint4* as = smem;
  int4* taps = reinterpret_cast<int4*>(smem + 8 * CELLS);
  const rfx_gemm_desc& d = g.d;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int P = d.OA * d.OB;
  // XCD-aware tile order (block b runs on XCD b % 8, each XCD has its own L2): the channel tiles of one
  // (sample, position tile) read the same input samples, so they are made consecutive ON THE SAME XCD;
  // neighbouring position tiles are spread over the 8 XCDs.
  const int mtiles = d.Mpad / BM, ptiles = (P + 127) / 128;
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int ym = q % mtiles;
  const int pw = g.xcd_chunk > 0 ? xcd * g.xcd_chunk + q / mtiles : (q / mtiles) * 8 + xcd;   // (n, position tile) work item
  if (pw >= ptiles * d.N) return;
  const int n = pw / ptiles;
  const int m0 = ym * BM;
  const int j = (pw - n * ptiles) * 128 + wave * 32 + l31;
  TileCtx tc;
  tc.n = n; tc.pw = pw; tc.m0 = m0; tc.wave = wave; tc.lane = lane; tc.l31 = l31; tc.h = h;
  tc.jvalid = j < P;
  const int jj = tc.jvalid ? j : 0;
  tc.a = jj / d.OB;
  tc.b = jj - tc.a * d.OB;
  TapLane c;
  const int ia0 = tc.a * d.SA, ib0 = tc.b * d.SB;
  c.ia0 = tc.jvalid ? ia0 : (1 << 30);
  c.ib0 = ib0;
  constexpr int ESZ = IN16 ? 2 : 4;                               // bytes per element of the gathered operand
  c.voff = (uint32_t)(((int64_t)ia0 * d.in_as + (int64_t)ib0 * d.in_bs) * ESZ);
  c.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(g.in) + (int64_t)n * d.in_ns * ESZ), 0,
                                           (int)d.in_extent, 0x00020000);
  c.t = 0; c.goff = 0;

  f32x16 acc[R];
#pragma unroll
  for (int mt = 0; mt < R; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  run_phase_tap<R, MODE, IN16>(d, g.apack, g.ktab, d.ntaps, d.gpt, d.Kpad_t, m0, c, as, taps, acc);
  fwd_epilogue_mid<R>(g, tc, acc);
  if (g.apack2 != nullptr) {
    if (g.in2) c.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(g.in2) + (int64_t)n * d.in_ns * ESZ), 0,
                                                        (int)d.in_extent, 0x00020000);
    run_phase_tap<R, MODE, IN16>(d, g.apack2, g.ktab2, g.ntaps2, d.gpt2 > 0 ? d.gpt2 : d.gpt, g.Kpad2, m0, c, as, taps, acc);
  }
  fwd_epilogue_store<R>(g, tc, acc);
}

// ---------------------------------------------------------------------------------
// Short reductions (Kpad_t <= 64: at most 4 K steps -- 1x1 convolutions over <= 64 channels, the DConv bottleneck pairs).
// These launches move far more bytes than they multiply and were LATENCY-bound in the tiled kernel above: a wave had one
// 32-position tile (4 KB of output at 32 rows) per memory round trip and only 8-16 waves fit a CU -> 2.3-2.8 TB/s (layer
// profile r02).  A first persistent version with the NEXT tile's gathers in flight did not help: hipcc drains
// vmcnt(0) at the loop header / at the register hand-over, so every iteration still paid one full round trip.
// This version raises the bytes per round trip instead: a wave owns NT consecutive 32-position tiles per iteration (all
// their gathers are issued back to back, then the MFMAs, then 16 * NT stores per lane), workgroups are persistent over
// position blocks (packed weights, tap table and bias are fetched once), one 32-row channel tile (R = 1).
// ---------------------------------------------------------------------------------
struct StreamGeo { int P, wave, lane, l31, h, m0, t0; uint32_t goff0, cs4, gstep, gwrap; int nk; };

template <int MODE>
__device__ __forceinline__ void stream_mma(const uint4* a_lds, int h, int l31, const float (&b)[8], f32x16& acc) {
  if (MODE == 1) {
    bf16x8 bh, bl;
    split8(b, bh, bl);
    const bf16x8 fh = __builtin_bit_cast(bf16x8, a_lds[h * 32 + l31]);
    const bf16x8 fl = __builtin_bit_cast(bf16x8, a_lds[64 + h * 32 + l31]);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl, bh, acc, 0, 0, 0);
  } else {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_lds[h * 32 + l31]), round8(b), acc, 0, 0, 0);
  }
}

__device__ __forceinline__ void stream_stat_flush(const FwdArgs& g, int n, int pw, int lane, double s1, double s2) {
  const double d1 = rfx_wave_sum_d(s1), d2 = rfx_wave_sum_d(s2);
  const int slots = g.e.stat_slots > 1 ? g.e.stat_slots : 1;
  double* dst = g.e.stat_sums + 2 * ((int64_t)n * slots + (pw & (slots - 1)));
  if (lane == 0) { atomicAdd(dst, d1); atomicAdd(dst + 1, d2); }
}

template <int MODE, int NT, int NKMAX>
__global__ __launch_bounds__(256, 2) void gemm_tap_stream_kernel(const FwdArgs g) {      // 3 waves / SIMD spills its tile ring: measured slower
  constexpr int R = 1, BM = 32, NARR = MODE == 1 ? 2 : 1, CELLS = 2 * NARR * BM;
  __shared__ __attribute__((aligned(16))) uint4 smem[4 * CELLS + RFX_TAP_LDS];   // A of K steps 0..3; tap table
  uint4* as = smem;
  int4* taps = reinterpret_cast<int4*>(smem + 4 * CELLS);
  const rfx_gemm_desc& d = g.d;
  const int tid = threadIdx.x;
  StreamGeo q;
  q.wave = tid >> 6; q.lane = tid & 63; q.l31 = q.lane & 31; q.h = q.lane >> 5;
  q.P = d.OA * d.OB;
  const int mtiles = d.Mpad / BM;
  const int ptiles = (q.P + 127) / 128;                          // 128-position tiles per sample
  const int tiles = ptiles * d.N;                                // a block of NT consecutive tiles may span samples
  const int work = (tiles + NT - 1) / NT;
  // the mtiles channel tiles of one position walk are blocks b, b + 8, b + 16, ... = the SAME XCD, started together and doing
  // equal work per tile: they read their (shared) input through one L2.  With `ym = b % mtiles` the channel tiles sat on
  // different XCDs and every one of them fetched the input from HBM (r02 per-launch PMC: reads = mtiles x the operand).
  const int xcd = blockIdx.x & 7, qq = blockIdx.x >> 3;
  const int ym = qq % mtiles, w0 = (qq / mtiles) * 8 + xcd, stride = gridDim.x / mtiles;
  q.m0 = ym * BM;
  q.nk = d.Kpad_t / 16;
  q.cs4 = (uint32_t)(d.in_cs * 4);
  q.gstep = 16u * q.cs4; q.gwrap = (uint32_t)d.gpt * 8u * q.cs4;
  {
    const uint4* apk = reinterpret_cast<const uint4*>(g.apack);
    const int64_t arr_stride = (int64_t)(d.Kpad_t / 8 + 8) * d.Mpad;
#pragma unroll
    for (int ks = 0; ks < NKMAX; ++ks)                           // the packed matrix carries >= 4 zero K steps of padding
      tap_a_store<R, MODE>(as + ks * CELLS, tid, tap_a_load<R, MODE>(apk, arr_stride, d.Mpad, 2 * ks, q.m0, tid));
    if (tid < d.ntaps + 16) taps[tid] = reinterpret_cast<const int4*>(g.ktab)[tid];
  }
  __syncthreads();
  q.t0 = q.h / d.gpt;                                            // group g = h of K step 0
  q.goff0 = (uint32_t)(q.h - q.t0 * d.gpt) * 8u * q.cs4;
  float bias[1][16];
  if (g.e.bias) fwd_load_bias<1>(g, q.m0, q.h, bias);

  const bool pair16 = d.out_bf16 != 0;                           // the launcher checked rfx_pair16_geo
  uint32_t frec; bool partial_ok;
  rfx_fast_store_geo(d, d.out_bf16 ? 2 : 4, &frec, &partial_ok);
  if (d.M % 32 == 0) frec = 0x7fffffffu;
  int stat_n = -1, stat_pw = 0;
  double st1 = 0., st2 = 0.;             // fp64 across tiles: how a sample's tiles group into iterations depends on the batch size, the sums must not
  for (int pw = w0; pw < work; pw += stride) {                   // wave-uniform trip count
    TileCtx tc[NT];
    float b[NT][NKMAX][8];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int gt = pw * NT + nt;                               // global tile index, wave-uniform
      const bool tv = gt < tiles;
      const int n = tv ? gt / ptiles : 0;
      const int j = (tv ? gt - n * ptiles : 0) * 128 + q.wave * 32 + q.l31;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.in + (int64_t)n * d.in_ns), 0,
                                                                          (int)d.in_extent, 0x00020000);
      tc[nt].n = n; tc[nt].pw = gt; tc[nt].m0 = q.m0; tc[nt].wave = q.wave; tc[nt].lane = q.lane;
      tc[nt].l31 = q.l31; tc[nt].h = q.h;
      tc[nt].jvalid = tv & (j < q.P);
      const int jj = tc[nt].jvalid ? j : 0;
      tc[nt].a = d.OA == 1 ? 0 : jj / d.OB;                     // wave-uniform select: the 1-D layers skip the division
      tc[nt].b = jj - tc[nt].a * d.OB;
      TapLane c;
      const int ia0 = tc[nt].a * d.SA, ib0 = tc[nt].b * d.SB;
      c.ia0 = tc[nt].jvalid ? ia0 : (1 << 30);
      c.ib0 = ib0;
      c.voff = (uint32_t)(((int64_t)ia0 * d.in_as + (int64_t)ib0 * d.in_bs) * 4);
      c.rs = rs;
      c.t = q.t0; c.goff = q.goff0;
#pragma unroll
      for (int ks = 0; ks < NKMAX; ++ks)
        if (ks == 0 || ks < q.nk) gather8_tap(d, taps, q.cs4, q.gstep, q.gwrap, c, b[nt][ks]);     // wave-uniform
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      f32x16 acc[1];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < NKMAX; ++ks)
        if (ks == 0 || ks < q.nk) stream_mma<MODE>(as + ks * CELLS, q.h, q.l31, b[nt][ks], acc[0]);
      if (g.e.bias) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] += bias[0][r];
      }
      // lean stores only: the launcher (rfx_tap_use_stream) sends every launch with a tile that could not take them to the tiled kernel.
      // 32 | P: a wave's 32 positions are valid or invalid together
      const bool tvalid = __builtin_amdgcn_ballot_w64(tc[nt].jvalid) != 0;
      float s1 = 0.f, s2 = 0.f;
      if (tvalid) {
        const int64_t opos = (int64_t)(tc[nt].a * d.out_sa + d.out_a0) * d.out_as + (int64_t)(tc[nt].b * d.out_sb + d.out_b0) * d.out_bs;
        if (g.e.glu_out) fwd_store_fast_glu<1>(g, tc[nt], acc, opos);
        else fwd_store_fast_plain<1>(g, tc[nt], acc, opos, pair16, frec, s1, s2);
      }
      if (g.e.stat_sums) {
        // GroupNorm(1, C) moments: summed per lane over the consecutive tiles of one sample, one wave reduction + fp64 atomic pair per run
        if (tvalid && tc[nt].n != stat_n) {
          if (stat_n >= 0) stream_stat_flush(g, stat_n, stat_pw, q.lane, st1, st2);
          stat_n = tc[nt].n; st1 = 0.; st2 = 0.;
        }
        if (tvalid) { st1 += (double)s1; st2 += (double)s2; stat_pw = tc[nt].pw; }
      }
    }
    if (g.e.stat_sums && stat_n >= 0) { stream_stat_flush(g, stat_n, stat_pw, q.lane, st1, st2); stat_n = -1; }
  }
}

// one translation unit per operand-storage variant (the tiled kernel is the slowest thing to compile in the library)
template <int IN16>
static int rfx_launch_gemm_tap_variant(const FwdArgs& g, int r, dim3 grid, hipStream_t s) {
  switch (r) {
    case 1: hipLaunchKernelGGL((gemm_tap_kernel<1, 2, IN16>), grid, dim3(256), 0, s, g); break;
    case 2: hipLaunchKernelGGL((gemm_tap_kernel<2, 2, IN16>), grid, dim3(256), 0, s, g); break;
    case 3: hipLaunchKernelGGL((gemm_tap_kernel<3, 2, IN16>), grid, dim3(256), 0, s, g); break;
    default: hipLaunchKernelGGL((gemm_tap_kernel<4, 2, IN16>), grid, dim3(256), 0, s, g); break;
  }
  RFX_CHECK_LAUNCH();
  return 0;
}
int rfx_launch_gemm_tap_in16(const FwdArgs& g, int r, dim3 grid, hipStream_t s);   // bf16 operand storage: gemm_fwd_bf16_in16.hip

template <int MODE>
static int rfx_launch_gemm_tap(const FwdArgs& g, int r, dim3 grid, hipStream_t s) {
  if (g.d.in_bf16 == 3) return -1;      // channels-last operand: template branch kept for the layout probe, not instantiated
  if (g.d.in_bf16) return (MODE == 2 && g.d.in_bf16 == 1) ? rfx_launch_gemm_tap_in16(g, r, grid, s) : -1;
  // short single-phase reductions with enough position tiles to keep persistent workgroups busy: streaming kernel
  if (rfx_tap_use_stream(g.d, g.e, g.apack2 != nullptr, r) && (!g.d.out_bf16 || (reinterpret_cast<uintptr_t>(g.out) & 3) == 0)) {
    const int mtiles = g.d.Mpad / 32;
    int nw = (512 / mtiles) & ~7;                                // persistent workgroups per channel tile (2 per CU in all),
    nw = nw < 8 ? 8 : nw;                                        // a multiple of 8: one walk per XCD slot (kernel's block order)
    dim3 sg((unsigned)(nw * mtiles));
    if (g.d.Kpad_t <= 16) hipLaunchKernelGGL((gemm_tap_stream_kernel<MODE, 4, 1>), sg, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_tap_stream_kernel<MODE, 2, 4>), sg, dim3(256), 0, s, g);
    RFX_CHECK_LAUNCH();
    return 0;
  }
  switch (r) {
    case 1: hipLaunchKernelGGL((gemm_tap_kernel<1, MODE>), grid, dim3(256), 0, s, g); break;
    case 2: hipLaunchKernelGGL((gemm_tap_kernel<2, MODE>), grid, dim3(256), 0, s, g); break;
    case 3: hipLaunchKernelGGL((gemm_tap_kernel<3, MODE>), grid, dim3(256), 0, s, g); break;
    default: hipLaunchKernelGGL((gemm_tap_kernel<4, MODE>), grid, dim3(256), 0, s, g); break;
  }
  RFX_CHECK_LAUNCH();
  return 0;
}
