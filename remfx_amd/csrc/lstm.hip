// LSTM recurrence (forward and backward through time) as ONE persistent launch per layer:
// the BiLSTM of HDemucs' DConv (torchaudio _BLSTM via models.py:319) and Open-Unmix's 3-layer
// BiLSTM (models.py:297-298).  Replaces cuDNN/MIOpen-style "one GEMM + one pointwise kernel per
// time step" (2624 GEMM launches per Demucs training step in the rocprof r01 traces).
//
// Everything around the recurrence is a plain gather-GEMM on channel-major tensors (1, C, T*Bn):
// input projections xp = W_ih x + b, and after the backward sweep dX, dW_ih, dW_hh, db.
// The sweep itself is a *wave cluster*:
//   * a cluster = (tile of 32 sequences, direction); it has H/16 single-wave workgroups, each owning 16 hidden
//     units (all four gates).  Every step a wave needs its slice of W_hh (4*16*H weights) and the whole h_{t-1};
//     one CU's vector L1 fills at 64 B/clk, so spreading a cluster's weights over H/16 CUs (register-resident in the
//     bf16 mode) is what makes a step cost ~ the exchange instead of ~ |W_hh| / 64 B/clk (measured: 30 us -> see DESIGN.md);
//   * gates^T[4H x 32] = xp[t] + W_hh[4H x H] . h^T[H x 32] on v_mfma_f32_32x32x16_bf16 with the
//     bf16x3 split (hi.hi + hi.lo + lo.hi, fp32 accumulate); MFMA rows = gate units, columns =
//     sequences, so every global access is coalesced along the sequence axis of the channel-major
//     tensors and a lane owns (8 units x 1 sequence) of all four gates -> the cell update is local;
//   * h_t (resp. the gate gradients in the backward sweep) is exchanged between the waves of a cluster through a
//     ping-pong buffer in global memory that is already in MFMA B-fragment order (bf16 hi / lo), written and
//     read with agent-scope atomic 8-byte accesses, plus one arrival counter per cluster;
//   * the grid never exceeds what is co-resident (<= RFX_LSTM_MAX_WAVES single-wave workgroups per launch, tiles are
//     chunked over launches), and every spin is bounded: on time-out the wave raises the error flag and exits.
//   Workgroup ids are laid out so that the waves of a cluster are congruent mod 8, i.e. on one XCD / one L2 when
//   the dispatcher round-robins (performance only; correctness relies on agent scope, not on placement).
#include "common.h"
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t lstm_bf16_rne(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
// 1-ulp hardware exp / rcp: absolute error ~1e-7 on values in [-1, 1], far inside the 1e-4 parity budget
__device__ __forceinline__ float lstm_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float lstm_tanh(float x) { return 2.f * __builtin_amdgcn_rcpf(1.f + __expf(-2.f * x)) - 1.f; }
__device__ __forceinline__ void split_hi_lo(float v, unsigned short& hi, unsigned short& lo) {
  const uint32_t u = __float_as_uint(v);
  const float r = v - __uint_as_float(u & 0xffff0000u);
  hi = (unsigned short)(u >> 16);
  lo = (unsigned short)((__float_as_uint(r) + 0x8000u) >> 16);
}

// fwdA[ub16][tl][ks][lane]: W_hh[(2tl + (m>>4))*H + 16ub16 + (m&15)][16ks + 8(lane>>5) + q], m = lane&31, q = 0..7 (hi array, then lo):
//   a forward wave owns 16 hidden units; its two 32-row MFMA tiles are (i | f) and (g | o) of those units, so that the four gates
//   of one (unit, sequence) meet in one lane (accumulator registers r and r + 8 of the two tiles)
// bwdA[ub][ks][lane]   : W_hh[16ks + 8(lane>>5) + q][32ub + (lane&31)]
__global__ void lstm_pack_kernel(const float* __restrict__ whh, int H, uint4* __restrict__ fwdA, uint4* __restrict__ bwdA) {
  const int nub = H / 32, nks = H / 16, nks4 = 4 * H / 16;
  const int64_t nf = (int64_t)nub * 4 * nks * 64, nb = (int64_t)nub * nks4 * 64;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < nf + nb; idx += (int64_t)gridDim.x * blockDim.x) {
    uint32_t hi[4] = {0, 0, 0, 0}, lo[4] = {0, 0, 0, 0};
    const bool f = idx < nf;
    const int64_t i = f ? idx : idx - nf;
    const int lane = (int)(i & 63);
    int64_t r = i >> 6;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float v;
      if (f) {
        const int ks = (int)(r % nks), tl = (int)((r / nks) % 2), ub = (int)(r / (2 * nks)), m = lane & 31;
        v = whh[(int64_t)((2 * tl + (m >> 4)) * H + 16 * ub + (m & 15)) * H + 16 * ks + 8 * (lane >> 5) + q];
      } else {
        const int ks = (int)(r % nks4), ub = (int)(r / nks4);
        v = whh[(int64_t)(16 * ks + 8 * (lane >> 5) + q) * H + 32 * ub + (lane & 31)];
      }
      const uint32_t h = lstm_bf16_rne(v);
      const uint32_t l = lstm_bf16_rne(v - __uint_as_float(h << 16));
      hi[q >> 1] |= h << (16 * (q & 1));
      lo[q >> 1] |= l << (16 * (q & 1));
    }
    uint4* dst = f ? fwdA : bwdA;
    const int64_t n = f ? nf : nb;
    dst[i] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    dst[n + i] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
  }
}


#define RFX_LSTM_MAX_WAVES 768    /* of 1024 one-wave-per-SIMD slots: leaves a quarter of the machine to concurrent kernels (RCCL) */
#define RFX_LSTM_SPIN_LIMIT (1 << 22)

struct LstmArgs {
  const float* xp;      // [2][4H][P]   input projections (+ both biases), P = T*Bn, position = t*Bn + b
  const uint4* packA;   // per direction: fwdA (hi, lo) then bwdA (hi, lo); dir stride in uint4 = pack_stride
  float* out;           // [2H][P]      channel-major output sequence (dir d at rows d*H..)
  float* gates;         // [2][4H][P]   post-activation i, f, g, o (NULL in inference)
  float* cstate;        // [2][H][P]
  const float* gout;    // bwd: [2H][P]
  float* dG;            // bwd: [2][4H][P] gate pre-activation gradients
  uint64_t* xch;        // exchange buffers, per cluster 2 x (KS k-steps x [hi|lo] x 64 lanes x 2) u64
  uint32_t* ctr;        // arrival counters, one per cluster, 64 B apart
  int32_t* err;         // sticky error flag (spin time-out)
  int64_t pack_stride;
  int T, Bn, H, P;
  int tile0, nclusters;
};

__device__ __forceinline__ bool lstm_wait(const uint32_t* ctr, uint32_t target) {
  for (int it = 0; it < RFX_LSTM_SPIN_LIMIT; ++it) {
    const uint32_t v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__builtin_amdgcn_readfirstlane(v) >= target) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}
// all of this wave's exchange stores have been acknowledged at agent scope -> publish
__device__ __forceinline__ void lstm_arrive(uint32_t* ctr, int lane) {
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): visible to the compiler's wait-count bookkeeping, unlike inline asm
  if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bf16x8 lstm_xch_load(const uint64_t* p) {
  const uint64_t a = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint64_t b = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint4 v = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
  return __builtin_bit_cast(bf16x8, v);
}
// four consecutive values -> bf16 hi / lo quads
__device__ __forceinline__ void lstm_xch_store(uint64_t* phi, uint64_t* plo, float v0, float v1, float v2, float v3) {
  unsigned short h0, h1, h2, h3, l0, l1, l2, l3;
  split_hi_lo(v0, h0, l0); split_hi_lo(v1, h1, l1); split_hi_lo(v2, h2, l2); split_hi_lo(v3, h3, l3);
  const uint64_t hi = (uint64_t)h0 | ((uint64_t)h1 << 16) | ((uint64_t)h2 << 32) | ((uint64_t)h3 << 48);
  const uint64_t lo = (uint64_t)l0 | ((uint64_t)l1 << 16) | ((uint64_t)l2 << 32) | ((uint64_t)l3 << 48);
  __hip_atomic_store(phi, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(plo, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// workgroup id -> (cluster, unit block): ids congruent mod 8 share an XCD under round-robin dispatch
__device__ __forceinline__ bool lstm_ids(const LstmArgs& a, int nwc, int& cluster, int& ub) {
  const int w = blockIdx.x, j = w >> 3;
  cluster = (j / nwc) * 8 + (w & 7);
  ub = j % nwc;
  return cluster < a.nclusters;
}

typedef uint32_t lstm_u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) lstm_u32x4* lstm_gptr_t;     // explicit global address space

struct LstmAFrag { uint4 h[2], l[2]; };

extern __shared__ __attribute__((aligned(16))) unsigned char lstm_smem[];

// n x (64 lanes x 16 B) from global (agent-coherent) straight into LDS, no registers: chunk i lands at lds + i KB
__device__ __forceinline__ void lstm_fetch_lds(const uint64_t* src, int n, int lane) {
  for (int i = 0; i < n; ++i)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (i * 64 + lane) * 2),
                                     (__attribute__((address_space(3))) void*)(lstm_smem + i * 1024), 16, 0,
                                     16 /* sc1: agent scope */);
}

// the hi halves only (LO = false kernels: the lo fragments are neither stored nor read): chunk 2 ks lands where loadB expects it
__device__ __forceinline__ void lstm_fetch_lds_hi(const uint64_t* src, int nks, int lane) {
  for (int ks = 0; ks < nks; ++ks)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (2 * ks * 64 + lane) * 2),
                                     (__attribute__((address_space(3))) void*)(lstm_smem + ks * 2048), 16, 0,
                                     16 /* sc1: agent scope */);
}
__device__ __forceinline__ void lstm_xch_store_hi(uint64_t* phi, float v0, float v1, float v2, float v3) {
  const uint64_t hi = (uint64_t)rfx_cvt_pk_bf16(v0, v1) | ((uint64_t)rfx_cvt_pk_bf16(v2, v3) << 32);
  __hip_atomic_store(phi, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Forward exchange (h_t as MFMA B fragments): u64 index ((ks*2 + part)*64 + L)*2 + half; L = (sequence, hhB) is the
// lane that consumes k = 16ks + 8hhB + 4half + e (e = 0..3, 16 bits each).  A forward wave owns the 16 units of ONE k-step (ks = its
// unit block ub): its lane (sequence l31, hh) holds units du = (r&3) + 8(r>>2) + 4hh for r = 0..7, i.e. for r = 4*hhB + e: lane
// l31+32hhB, half hh.  (16 units per wave, not 32: the cell update -- 5 quarter-rate transcendentals per (unit, sequence) -- was
// 1.9 us of a 5 us step on the one wave a SIMD runs; the machine has 1024 such slots and a layer fills < 100 of them.)
// D = depth of the weight-fragment ring in k-steps (each k-step = 2 tiles x (hi, lo) = 4 KB per wave): the L2 round trip
// (~0.5 us) is covered only if several k-steps are in flight; D == NKS: the wave's whole W_hh slice stays in registers.
// NKS = H/16 at compile time (0 = runtime trip count).  With a runtime k loop hipcc drains vmcnt(0) at the loop header
// (55 instead of 32 cycles per MFMA, in-kernel s_memtime stamps).  Fully unrolled, its wait counts are exact
// -- provided the fragment base pointer is made opaque once per time step (otherwise ~100 hoisted
// addresses spill) and keeps its global address space (a laundered generic pointer turns every load into flat_load).
// LO = true: bf16x3 split products (hi.hi + hi.lo + lo.hi); false: bf16 operands only (the bf16-mixed mode: a third of the
// MFMA chain and half of the weight stream per step).
template <int D, int NKS, bool LO>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_fwd_kernel(const LstmArgs a) {
  int cluster, ub;
  if (!lstm_ids(a, a.H / 16, cluster, ub)) return;
  const int H = a.H, Bn = a.Bn, T = a.T, P = a.P;
  const int dir = cluster & 1, row0 = (a.tile0 + (cluster >> 1)) * 32;
  const int lane = threadIdx.x, l31 = lane & 31, hh = lane >> 5;
  const int nks = NKS ? NKS : H / 16, nwc = nks;                       // waves per cluster = unit blocks of 16
  const int row = row0 + l31;
  const bool rvalid = row < Bn;
  const int rowc = rvalid ? row : 0;      // idle lanes read sequence 0 and never store: every load stays unconditional
  const float* __restrict__ xp = a.xp + (int64_t)dir * 4 * H * P;
  const int nf = nwc * 2 * nks * 64;
  uint64_t abase = (uint64_t)(a.packA + (int64_t)dir * a.pack_stride + ub * 2 * nks * 64);   // wave-uniform
  lstm_gptr_t Ahi = (lstm_gptr_t)abase;
  lstm_gptr_t Alo = Ahi + nf;
  float* outp = a.out + (int64_t)dir * H * P;
  float* gsave = a.gates ? a.gates + (int64_t)dir * 4 * H * P : nullptr;
  float* csave = a.gates ? a.cstate + (int64_t)dir * H * P : nullptr;
  const int bufw = nks * 256;                                        // u64 per ping-pong buffer
  uint64_t* xch = a.xch + (int64_t)cluster * 2 * bufw;
  uint32_t* ctr = a.ctr + cluster * 16;
  const uint32_t uP = (uint32_t)P, HP = (uint32_t)H * uP;
  const uint32_t ubase = (uint32_t)(16 * ub + 4 * hh) * uP + (uint32_t)rowc;   // + ((r&3) + 8(r>>2)) * P + t*Bn, r < 8
  const uint32_t lane16 = (uint32_t)lane * 16u;
  auto loadA = [&](LstmAFrag& f, int ks) {
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
      // wave-uniform base + zero-extended lane offset -> scalar-base addressing (no per-fragment 64-bit VGPR address)
      f.h[tl] = __builtin_bit_cast(uint4, *(lstm_gptr_t)((uint64_t)(Ahi + (tl * nks + ks) * 64) + lane16));
      if (LO) f.l[tl] = __builtin_bit_cast(uint4, *(lstm_gptr_t)((uint64_t)(Alo + (tl * nks + ks) * 64) + lane16));
    }
  };
  auto loadB = [&](bf16x8& bh, bf16x8& bl, int ks) {
    bh = *reinterpret_cast<const bf16x8*>(lstm_smem + ks * 2048 + lane * 16);
    if (LO) bl = *reinterpret_cast<const bf16x8*>(lstm_smem + ks * 2048 + 1024 + lane * 16);
  };
  // the two tile accumulators are independent: alternate them so that no MFMA waits on its predecessor
  auto mma = [&](f32x16* acc, const LstmAFrag& f, const bf16x8 bh, const bf16x8 bl) {
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
      acc[tl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.h[tl]), bh, acc[tl], 0, 0, 0);
    if (LO) {
#pragma unroll
      for (int tl = 0; tl < 2; ++tl)
        acc[tl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.h[tl]), bl, acc[tl], 0, 0, 0);
#pragma unroll
      for (int tl = 0; tl < 2; ++tl)
        acc[tl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.l[tl]), bh, acc[tl], 0, 0, 0);
    }
  };
  float c[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) c[r] = 0.f;
  f32x16 acc[2];
  LstmAFrag f[D];
#pragma unroll
  for (int d = 0; d < D; ++d) loadA(f[d], d);
  // xpv = this step's input projections: fetched at the END of the previous step (after the publish), so that the loads and their
  // address arithmetic run while h_t is in flight instead of between the fetch and the MFMA chain
  float xpv[4][8];
  {
    const uint32_t o0 = ubase + (uint32_t)((dir == 0 ? 0 : T - 1) * Bn);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 8; ++r) xpv[g][r] = xp[o0 + (uint32_t)g * HP + (uint32_t)((r & 3) + 8 * (r >> 2)) * uP];
  }
  for (int s = 0; s < T; ++s) {
    if (NKS) { asm volatile("" : "+s"(abase)); Ahi = (lstm_gptr_t)abase; Alo = Ahi + nf; }   // opaque base, see above

    const int t = dir == 0 ? s : T - 1 - s;
    const uint32_t o0 = ubase + (uint32_t)(t * Bn);
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tl][r] = 0.f;
    if (s > 0) {
      if (!lstm_wait(ctr, (uint32_t)(nwc * s))) { *a.err = 1; return; }
      if (LO) lstm_fetch_lds(xch + ((s - 1) & 1) * bufw, 2 * nks, lane);
      else lstm_fetch_lds_hi(xch + ((s - 1) & 1) * bufw, nks, lane);          // half the exchange volume in the bf16 mode
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): visible to the compiler's wait-count bookkeeping, unlike inline asm
    }
    if (s > 0) {
      bf16x8 bh[2], bl[2];
      loadB(bh[0], bl[0], 0);
      auto kstep = [&](int ks) {                    // D k-steps: MFMAs of slot d, then its refill D k-steps ahead
#pragma unroll
        for (int d = 0; d < D; ++d) {
          const int k1 = ks + d + 1 == nks ? 0 : ks + d + 1;       // h fragments one k-step ahead (LDS latency)
          loadB(bh[(d + 1) & 1], bl[(d + 1) & 1], k1);
          mma(acc, f[d], bh[d & 1], bl[d & 1]);
          if (!(NKS && D == NKS)) {                   // D == NKS: the whole W_hh slice of this wave stays in registers
            const int kn = ks + d + D;                // the slot's next occupant; wraps into the next step
            loadA(f[d], kn >= nks ? kn - nks : kn);
          }
          __builtin_amdgcn_sched_barrier(0);          // keep the refill behind its MFMAs (scheduler would hoist all loads)
        }
      };
      if (NKS) {
#pragma unroll
        for (int ks = 0; ks < NKS; ks += D) kstep(ks);
      } else {
        for (int ks = 0; ks < nks; ks += D) kstep(ks);              // nks % D == 0, D even
      }
    }
    uint64_t* hw = xch + (s & 1) * bufw;
    float hv[8], gi[8], gf[8], gc[8], go[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float ig = lstm_sigmoid(acc[0][r] + xpv[0][r]), fg = lstm_sigmoid(acc[0][r + 8] + xpv[1][r]);
      const float gg = lstm_tanh(acc[1][r] + xpv[2][r]), og = lstm_sigmoid(acc[1][r + 8] + xpv[3][r]);
      c[r] = fg * c[r] + ig * gg;
      hv[r] = og * lstm_tanh(c[r]);
      gi[r] = ig; gf[r] = fg; gc[r] = gg; go[r] = og;
    }
    if (s + 1 < T) {                  // publish h_t first: it is the cluster's critical path
#pragma unroll
      for (int hb2 = 0; hb2 < 2; ++hb2) {
        const int r0 = 4 * hb2;
        uint64_t* d = hw + ((ub * 2) * 64 + l31 + 32 * hb2) * 2 + hh;
        if (LO) lstm_xch_store(d, d + 128, hv[r0], hv[r0 + 1], hv[r0 + 2], hv[r0 + 3]);
        else lstm_xch_store_hi(d, hv[r0], hv[r0 + 1], hv[r0 + 2], hv[r0 + 3]);
      }
      lstm_arrive(ctr, lane);
    }
    if (rvalid) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const uint32_t o = o0 + (uint32_t)((r & 3) + 8 * (r >> 2)) * uP;
        outp[o] = hv[r];
        if (gsave) {
          gsave[o] = gi[r]; gsave[o + HP] = gf[r]; gsave[o + 2 * HP] = gc[r]; gsave[o + 3 * HP] = go[r];
          csave[o] = c[r];
        }
      }
    }
    if (s + 1 < T) {
      const uint32_t on = ubase + (uint32_t)((dir == 0 ? s + 1 : T - 2 - s) * Bn);
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 8; ++r) xpv[g][r] = xp[on + (uint32_t)g * HP + (uint32_t)((r & 3) + 8 * (r >> 2)) * uP];
    }
  }
}

// Backward through time.  A wave owns 16 hidden units (as in the forward sweep): it multiplies ITS gate-gradient slice
// (K = 4 gates x 16 rows of W_hh) into partial dh_{t-1} for ALL hidden units and publishes the fp32 partial tiles; the owner of
// a unit block sums the H/16 partials.  Exchange (per ping-pong buffer): [consumer unit block][producer][quad jj][lane][4 floats]
// -- a 32-unit MFMA tile's accumulator quads j = 0, 1 are its first 16 units, j = 2, 3 the second 16 -- i.e. the consumer's
// slice is H/16 x 2 KB, contiguous, fetched straight into LDS.
// TPC = output tiles per round (independent accumulators); RES: the wave's whole W_hh^T slice (4 fragments per output tile) stays
// in registers; otherwise a ring of 4*TPC fragments is refilled behind its MFMAs.
template <int TPC, int NWC, bool LO, bool RES>   // NWC = H/32 at compile time (0 = runtime), as NKS in the forward kernel; LO as there
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_bwd_kernel(const LstmArgs a) {
  static_assert(!RES || NWC > 0, "resident weights need the compile-time tile count");
  int cluster, ub;
  if (!lstm_ids(a, a.H / 16, cluster, ub)) return;
  const int H = a.H, Bn = a.Bn, T = a.T, P = a.P;
  const int dir = cluster & 1, row0 = (a.tile0 + (cluster >> 1)) * 32;
  const int lane = threadIdx.x, l31 = lane & 31, hh = lane >> 5;
  const int nwc = NWC ? NWC : H / 32, nwv = 2 * nwc, nks = nwv, nks4 = 4 * nks;    // output tiles, waves per cluster, k-steps
  const int row = row0 + l31;
  const bool rvalid = row < Bn;
  const int rowc = rvalid ? row : 0;
  const int nf = nwc * 4 * nks * 64, nb = nwc * nks4 * 64;
  // bwdA[ot][ks'][lane]: tile ot = output unit block of 32, ks' = gate-row k-step; this wave's rows: g*nks + ub
  uint64_t abase = (uint64_t)(a.packA + (int64_t)dir * a.pack_stride + 2 * nf + ub * 64);   // wave-uniform
  lstm_gptr_t Ahi = (lstm_gptr_t)abase;
  lstm_gptr_t Alo = Ahi + nb;
  const float* __restrict__ gates = a.gates + (int64_t)dir * 4 * H * P;
  const float* __restrict__ cst = a.cstate + (int64_t)dir * H * P;
  float* __restrict__ dG = a.dG + (int64_t)dir * 4 * H * P;
  const float* __restrict__ goutp = a.gout + (int64_t)dir * H * P;
  const int bufw = nwv * nwv * 256;                                  // u64 per ping-pong buffer
  uint64_t* xch = a.xch + (int64_t)cluster * 2 * bufw;
  uint32_t* ctr = a.ctr + cluster * 16;
  const uint32_t uP = (uint32_t)P, HP = (uint32_t)H * uP;
  const uint32_t ubase = (uint32_t)(16 * ub + 4 * hh) * uP + (uint32_t)rowc;
  constexpr int RING = RES ? 4 * NWC : 4 * TPC;
  const int F = 4 * nwc;                                             // fragments per time step, F % RING == 0
  auto frag_index = [&](int fi) { const int ot = fi >> 2, g = fi & 3; return (ot * nks4 + g * nks) * 64 + lane; };
  uint4 wh[RING], wl[LO ? RING : 1];
#pragma unroll
  for (int i = 0; i < RING; ++i) {
    wh[i] = __builtin_bit_cast(uint4, Ahi[frag_index(i)]);
    if (LO) wl[i] = __builtin_bit_cast(uint4, Alo[frag_index(i)]);
  }
  float dcc[8], ct[8];
  float sg[4][8], cp[8], gy[8];            // this step's saved gates / c_{prev} / incoming gradient: fetched at the end of the previous step
  {
    const uint32_t o0 = ubase + (uint32_t)((dir == 0 ? T - 1 : 0) * Bn);
    const uint32_t op0 = ubase + (uint32_t)((T > 1 ? (dir == 0 ? T - 2 : 1) : (dir == 0 ? T - 1 : 0)) * Bn);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const uint32_t du = (uint32_t)((r & 3) + 8 * (r >> 2)) * uP;
      dcc[r] = 0.f;
      ct[r] = cst[o0 + du];
      sg[0][r] = gates[o0 + du]; sg[1][r] = gates[o0 + du + HP]; sg[2][r] = gates[o0 + du + 2 * HP]; sg[3][r] = gates[o0 + du + 3 * HP];
      cp[r] = cst[op0 + du];
      gy[r] = goutp[o0 + du];
    }
  }
  for (int s = T - 1; s >= 0; --s) {            // reverse of the forward processing order
    if (NWC && !RES) { asm volatile("" : "+s"(abase)); Ahi = (lstm_gptr_t)abase; Alo = Ahi + nb; }   // opaque base (see the forward kernel)
    const int step = T - 1 - s;                 // exchange generation
    const int t = dir == 0 ? s : T - 1 - s;
    const uint32_t o0 = ubase + (uint32_t)(t * Bn);
    if (step > 0) {
      if (!lstm_wait(ctr, (uint32_t)(nwv * step))) { *a.err = 1; return; }
      lstm_fetch_lds(xch + ((step - 1) & 1) * bufw + ub * nwv * 256, 2 * nwv, lane);
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): visible to the compiler's wait-count bookkeeping, unlike inline asm
    }
    if (step > 0) {
      for (int p = 0; p < nwv; ++p)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(lstm_smem + (p * 2 + jj) * 1024 + lane * 16);
          gy[4 * jj] += v[0]; gy[4 * jj + 1] += v[1]; gy[4 * jj + 2] += v[2]; gy[4 * jj + 3] += v[3];
        }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float ig = sg[0][r], fg = sg[1][r], gg = sg[2][r], og = sg[3][r];
      const float cprev = s > 0 ? cp[r] : 0.f;
      const float th = lstm_tanh(ct[r]);
      const float dc = gy[r] * og * (1.f - th * th) + dcc[r];
      sg[3][r] = gy[r] * th * og * (1.f - og);
      sg[0][r] = dc * gg * ig * (1.f - ig);
      sg[1][r] = dc * cprev * fg * (1.f - fg);
      sg[2][r] = dc * ig * (1.f - gg * gg);
      dcc[r] = dc * fg;
      ct[r] = cp[r];
    }
    if (s > 0) {
      // own gate gradients as MFMA B fragments: lane (seq, hhB) needs units 8hhB + q; q < 4 live in the
      // hh = 0 lane, q >= 4 in the hh = 1 lane of the same sequence -> one cross-half exchange per packed pair
      uint4 bh[4], bl[LO ? 4 : 1];                           // index g
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint32_t x0h[2], x0l[2], x1h[2], x1l[2];             // packed bf16 pairs of X0 = r 0..3, X1 = r 4..7
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          if (LO) {
            unsigned short h0, l0, h1, l1;
            split_hi_lo(sg[g][2 * e], h0, l0); split_hi_lo(sg[g][2 * e + 1], h1, l1);
            x0h[e] = (uint32_t)h0 | ((uint32_t)h1 << 16); x0l[e] = (uint32_t)l0 | ((uint32_t)l1 << 16);
            split_hi_lo(sg[g][4 + 2 * e], h0, l0); split_hi_lo(sg[g][4 + 2 * e + 1], h1, l1);
            x1h[e] = (uint32_t)h0 | ((uint32_t)h1 << 16); x1l[e] = (uint32_t)l0 | ((uint32_t)l1 << 16);
          } else {
            x0h[e] = rfx_cvt_pk_bf16(sg[g][2 * e], sg[g][2 * e + 1]);
            x1h[e] = rfx_cvt_pk_bf16(sg[g][4 + 2 * e], sg[g][4 + 2 * e + 1]);
            x0l[e] = x1l[e] = 0;
          }
        }
        uint32_t rh[2], rl[2] = {0, 0};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          rh[e] = (uint32_t)__shfl_xor((int)(hh ? x0h[e] : x1h[e]), 32, 64);
          if (LO) rl[e] = (uint32_t)__shfl_xor((int)(hh ? x0l[e] : x1l[e]), 32, 64);
        }
        bh[g] = hh ? make_uint4(rh[0], rh[1], x1h[0], x1h[1]) : make_uint4(x0h[0], x0h[1], rh[0], rh[1]);
        if (LO) bl[g] = hh ? make_uint4(rl[0], rl[1], x1l[0], x1l[1]) : make_uint4(x0l[0], x0l[1], rl[0], rl[1]);
      }
      uint64_t* gw = xch + (step & 1) * bufw;
      auto round = [&](int f0) {
        f32x16 acc[TPC];
#pragma unroll
        for (int tp = 0; tp < TPC; ++tp)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[tp][r] = 0.f;
        // the TPC tiles of a round are independent accumulators: interleave them term by term
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const bf16x8 Bh = __builtin_bit_cast(bf16x8, bh[g]), Bl = __builtin_bit_cast(bf16x8, bl[LO ? g : 0]);
#pragma unroll
          for (int tp = 0; tp < TPC; ++tp)
            acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wh[(RES ? f0 : 0) + tp * 4 + g]), Bh, acc[tp], 0, 0, 0);
          if (LO) {
#pragma unroll
            for (int tp = 0; tp < TPC; ++tp)
              acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wh[(RES ? f0 : 0) + tp * 4 + g]), Bl, acc[tp], 0, 0, 0);
#pragma unroll
            for (int tp = 0; tp < TPC; ++tp)
              acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wl[LO ? (RES ? f0 : 0) + tp * 4 + g : 0]), Bh, acc[tp], 0, 0, 0);
          }
          if (!RES) {
#pragma unroll
            for (int tp = 0; tp < TPC; ++tp) {
              const int i = tp * 4 + g;
              int fn = f0 + i + RING;                                 // the slot's next occupant (wraps into the next step)
              fn = fn >= F ? fn - F : fn;
              wh[i] = __builtin_bit_cast(uint4, Ahi[frag_index(fn)]);
              if (LO) wl[LO ? i : 0] = __builtin_bit_cast(uint4, Alo[frag_index(fn)]);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
#pragma unroll
        for (int tp = 0; tp < TPC; ++tp) {
          const int ot = (f0 >> 2) + tp;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            // quads 0, 1 -> consumer 2 ot (its units 0..15 of the tile), quads 2, 3 -> consumer 2 ot + 1
            uint64_t* d = gw + (int64_t)(((2 * ot + (j >> 1)) * nwv + ub) * 2 + (j & 1)) * 128 + lane * 2;
            const uint64_t v0 = (uint64_t)__float_as_uint(acc[tp][4 * j]) | ((uint64_t)__float_as_uint(acc[tp][4 * j + 1]) << 32);
            const uint64_t v1 = (uint64_t)__float_as_uint(acc[tp][4 * j + 2]) | ((uint64_t)__float_as_uint(acc[tp][4 * j + 3]) << 32);
            __hip_atomic_store(d, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(d + 1, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      };
      if (NWC) {
#pragma unroll
        for (int f0 = 0; f0 < 4 * NWC; f0 += 4 * TPC) round(f0);
      } else {
        for (int f0 = 0; f0 < F; f0 += 4 * TPC) round(f0);
      }
      lstm_arrive(ctr, lane);
    }
    if (rvalid) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const uint32_t o = o0 + (uint32_t)((r & 3) + 8 * (r >> 2)) * uP;
        dG[o] = sg[0][r]; dG[o + HP] = sg[1][r]; dG[o + 2 * HP] = sg[2][r]; dG[o + 3 * HP] = sg[3][r];
      }
    }
    if (s > 0) {
      // next step (s-1): its time index and the one before it in forward order (clamped at the sequence start); issued behind the
      // publish so that the loads and their address arithmetic run while the partial tiles are in flight
      const int s1 = s - 1, s2 = s1 > 0 ? s1 - 1 : 0;
      const uint32_t on = ubase + (uint32_t)((dir == 0 ? s1 : T - 1 - s1) * Bn);
      const uint32_t opn = ubase + (uint32_t)((dir == 0 ? s2 : T - 1 - s2) * Bn);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const uint32_t du = (uint32_t)((r & 3) + 8 * (r >> 2)) * uP;
        sg[0][r] = gates[on + du]; sg[1][r] = gates[on + du + HP]; sg[2][r] = gates[on + du + 2 * HP]; sg[3][r] = gates[on + du + 3 * HP];
        cp[r] = cst[opn + du];
        gy[r] = goutp[on + du];
      }
    }
  }
}

// ---- single-workgroup clusters (H <= 192, bf16 operands) ---------------------------------------------------------------------
// The cluster form above spreads a cluster's W_hh over H/16 CUs and pays three L2 round trips per time step for it (publish, arrive,
// fetch: ~1.5 of the 2.3 us of a forward step at H = 192, and 295 KB of fp32 partial tiles per backward step).  At H = 192 the whole
// bf16 W_hh (295 KB) fits ONE CU's register file: a cluster = (tile of 16 sequences, direction) is then ONE workgroup of H/16 waves,
// each wave keeps its 16 units' rows of W_hh (forward) resp. columns (backward) in 96 registers, and h_t / the gate gradients are
// exchanged through LDS with one barrier per step.  Sequences are independent, so the batch is simply split over CUs (16 sequences
// each: v_mfma_f32_16x16x32_bf16, rows = this wave's units, columns = sequences) -- no cross-CU traffic at all, no co-residency rule.
// Backward: every wave contracts ALL gate gradients (K = 4 H, read from LDS) with its own units' columns of W_hh, so dh_{t-1} of its
// units stays in its registers: one exchange per step (48 KB of bf16 gate gradients per cluster in LDS) instead of the partial tiles.
// Lane l: sequence l & 15, units 4 (l >> 4) + r (r = 0..3) of the wave's 16 -- the C/D layout of the 16 x 16 MFMA, so the four gates
// of a (unit, sequence) meet in one lane (four accumulators, one per gate) and the cell update is local.
typedef float lstm_f32x4 __attribute__((ext_vector_type(4)));

// localA: per direction [fwd | bwd]; fwd[w][g][ks][lane]: W_hh[g H + 16 w + (lane & 15)][32 ks + 8 (lane >> 4) + j];
//                                    bwd[w][ks][lane]:    W_hh[32 ks + 8 (lane >> 4) + j][16 w + (lane & 15)]   (j = 0..7, bf16)
__global__ void lstm_pack_local_kernel(const float* __restrict__ whh, int H, uint4* __restrict__ dst) {
  const int nw = H / 16, nk = H / 32, nk4 = 4 * nk;
  const int64_t nf = (int64_t)nw * 4 * nk * 64, nb = (int64_t)nw * nk4 * 64;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < nf + nb; idx += (int64_t)gridDim.x * blockDim.x) {
    const bool f = idx < nf;
    const int64_t i = f ? idx : idx - nf;
    const int lane = (int)(i & 63), row = lane & 15, kb = lane >> 4;
    int64_t r = i >> 6;
    uint32_t pk[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v;
      if (f) {
        const int ks = (int)(r % nk), g = (int)((r / nk) % 4), w = (int)(r / (4 * nk));
        v = whh[(int64_t)(g * H + 16 * w + row) * H + 32 * ks + 8 * kb + j];
      } else {
        const int ks = (int)(r % nk4), w = (int)(r / nk4);
        v = whh[(int64_t)(32 * ks + 8 * kb + j) * H + 16 * w + row];
      }
      pk[j >> 1] |= lstm_bf16_rne(v) << (16 * (j & 1));
    }
    dst[idx] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
  }
}
static int64_t lstm_local_uint4_per_dir(int H) { return (H <= 192) ? (int64_t)2 * (H / 16) * 4 * (H / 32) * 64 : 0; }

// h (or a gate gradient) of this lane's four units -> the B fragment slot its consumers read: k = unit index within the operand
__device__ __forceinline__ void lstm_local_publish(unsigned char* buf, int k0, int sq, float v0, float v1, float v2, float v3) {
  // k = k0 .. k0 + 3 (k0 % 4 == 0): k-step k / 32, fragment lane 16 ((k % 32) / 8) + sequence, bytes 2 (k % 8) ..
  const int ks = k0 >> 5, kk = k0 & 31;
  *reinterpret_cast<uint2*>(buf + ks * 1024 + (16 * (kk >> 3) + sq) * 16 + 2 * (kk & 7)) =
      make_uint2(rfx_cvt_pk_bf16(v0, v1), rfx_cvt_pk_bf16(v2, v3));
}
__device__ __forceinline__ void lstm_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// (The other MFMA orientation -- rows = sequences, a lane = one unit at four consecutive sequences, 16-byte accesses -- was built
// too: a quarter of the memory instructions, but every lane of an instruction then touches its own cache line (units are P floats
// apart): forward step 2.4 -> 3.9 us.  Lanes along the sequences keep the 64-byte runs.)

template <int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(3, 3))) void lstm_fwd_local_kernel(const LstmArgs a, const uint4* __restrict__ lpack,
                                                                                                             int64_t lstride) {
  constexpr int H = 16 * NW, NK = H / 32;
  const int cluster = blockIdx.x, dir = cluster & 1, tile = cluster >> 1;
  const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, sq = lane & 15, qb = lane >> 4;
  const int Bn = a.Bn, T = a.T, P = a.P;
  const int b = tile * 16 + sq;
  const bool rvalid = b < Bn;
  const int bc = rvalid ? b : 0;          // idle lanes read sequence 0 and never store
  const float* __restrict__ xp = a.xp + (int64_t)dir * 4 * H * P;
  float* outp = a.out + (int64_t)dir * H * P;
  float* gsave = a.gates ? a.gates + (int64_t)dir * 4 * H * P : nullptr;
  float* csave = a.gates ? a.cstate + (int64_t)dir * H * P : nullptr;
  const uint32_t uP = (uint32_t)P, HP = (uint32_t)H * uP;
  const int u0 = 16 * w + 4 * qb;                                   // first of this lane's four units
  const uint32_t ubase = (uint32_t)u0 * uP + (uint32_t)bc;
  // this wave's rows of W_hh, all four gates: 4 NK fragments = 96 registers at H = 192
  uint4 wf[4][NK];
  {
    const uint4* A = lpack + (int64_t)dir * lstride + (int64_t)w * 4 * NK * 64 + lane;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) wf[g][ks] = A[(g * NK + ks) * 64];
  }
  float c[4] = {0.f, 0.f, 0.f, 0.f};
  float xpv[4][4];
  {
    const uint32_t o0 = ubase + (uint32_t)((dir == 0 ? 0 : T - 1) * Bn);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) xpv[g][r] = xp[o0 + (uint32_t)g * HP + (uint32_t)r * uP];
  }
  for (int s = 0; s < T; ++s) {
    const int t = dir == 0 ? s : T - 1 - s;
    const uint32_t o0 = ubase + (uint32_t)(t * Bn);
    lstm_f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = lstm_f32x4{0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
      const unsigned char* hb = lstm_smem + ((s - 1) & 1) * (NK * 1024) + lane * 16;
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) {
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(hb + ks * 1024);
#pragma unroll
        for (int g = 0; g < 4; ++g)
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[g][ks]), bh, acc[g], 0, 0, 0);
      }
    }
    float hv[4], gi[4], gf[4], gc[4], go[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float ig = lstm_sigmoid(acc[0][r] + xpv[0][r]), fg = lstm_sigmoid(acc[1][r] + xpv[1][r]);
      const float gg = lstm_tanh(acc[2][r] + xpv[2][r]), og = lstm_sigmoid(acc[3][r] + xpv[3][r]);
      c[r] = fg * c[r] + ig * gg;
      hv[r] = og * lstm_tanh(c[r]);
      gi[r] = ig; gf[r] = fg; gc[r] = gg; go[r] = og;
    }
    if (s + 1 < T) {                     // h_t for everybody's next step: LDS, one barrier
      lstm_local_publish(lstm_smem + (s & 1) * (NK * 1024), u0, sq, hv[0], hv[1], hv[2], hv[3]);
      // the next step's input projections fly under the barrier and the MFMA chain
      const uint32_t on = ubase + (uint32_t)((dir == 0 ? s + 1 : T - 2 - s) * Bn);
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) xpv[g][r] = xp[on + (uint32_t)g * HP + (uint32_t)r * uP];
      lstm_lds_barrier();
    }
    if (rvalid) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t o = o0 + (uint32_t)r * uP;
        outp[o] = hv[r];
        if (gsave) {
          gsave[o] = gi[r]; gsave[o + HP] = gf[r]; gsave[o + 2 * HP] = gc[r]; gsave[o + 3 * HP] = go[r];
          csave[o] = c[r];
        }
      }
    }
  }
}

template <int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(3, 3))) void lstm_bwd_local_kernel(const LstmArgs a, const uint4* __restrict__ lpack,
                                                                                                             int64_t lstride) {
  constexpr int H = 16 * NW, NK = H / 32, NK4 = 4 * NK;
  const int cluster = blockIdx.x, dir = cluster & 1, tile = cluster >> 1;
  const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, sq = lane & 15, qb = lane >> 4;
  const int Bn = a.Bn, T = a.T, P = a.P;
  const int b = tile * 16 + sq;
  const bool rvalid = b < Bn;
  const int bc = rvalid ? b : 0;
  const float* __restrict__ gates = a.gates + (int64_t)dir * 4 * H * P;
  const float* __restrict__ cst = a.cstate + (int64_t)dir * H * P;
  float* __restrict__ dG = a.dG + (int64_t)dir * 4 * H * P;
  const float* __restrict__ goutp = a.gout + (int64_t)dir * H * P;
  const uint32_t uP = (uint32_t)P, HP = (uint32_t)H * uP;
  const int u0 = 16 * w + 4 * qb;
  const uint32_t ubase = (uint32_t)u0 * uP + (uint32_t)bc;
  // this wave's columns of W_hh (rows of W_hh^T) over all 4 H gate rows: NK4 fragments = 96 registers at H = 192
  uint4 wb[NK4];
  {
    const uint4* A = lpack + (int64_t)dir * lstride + (int64_t)NW * 4 * NK * 64 + (int64_t)w * NK4 * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < NK4; ++ks) wb[ks] = A[ks * 64];
  }
  float dcc[4] = {0.f, 0.f, 0.f, 0.f}, ct[4], sg[4][4], cp[4], gy[4];
  {
    const uint32_t o0 = ubase + (uint32_t)((dir == 0 ? T - 1 : 0) * Bn);
    const uint32_t op0 = ubase + (uint32_t)((T > 1 ? (dir == 0 ? T - 2 : 1) : (dir == 0 ? T - 1 : 0)) * Bn);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t du = (uint32_t)r * uP;
      ct[r] = cst[o0 + du];
      sg[0][r] = gates[o0 + du]; sg[1][r] = gates[o0 + du + HP]; sg[2][r] = gates[o0 + du + 2 * HP]; sg[3][r] = gates[o0 + du + 3 * HP];
      cp[r] = cst[op0 + du];
      gy[r] = goutp[o0 + du];
    }
  }
  lstm_f32x4 rec = lstm_f32x4{0.f, 0.f, 0.f, 0.f};          // W_hh^T . (gate gradients of the step before): dh of this lane's units
  for (int s = T - 1; s >= 0; --s) {                          // reverse of the forward processing order
    const int step = T - 1 - s;
    const int t = dir == 0 ? s : T - 1 - s;
    const uint32_t o0 = ubase + (uint32_t)(t * Bn);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float dh = gy[r] + rec[r];
      const float ig = sg[0][r], fg = sg[1][r], gg = sg[2][r], og = sg[3][r];
      const float cprev = s > 0 ? cp[r] : 0.f;
      const float th = lstm_tanh(ct[r]);
      const float dc = dh * og * (1.f - th * th) + dcc[r];
      sg[3][r] = dh * th * og * (1.f - og);
      sg[0][r] = dc * gg * ig * (1.f - ig);
      sg[1][r] = dc * cprev * fg * (1.f - fg);
      sg[2][r] = dc * ig * (1.f - gg * gg);
      dcc[r] = dc * fg;
      ct[r] = cp[r];
    }
    unsigned char* gb = lstm_smem + (step & 1) * (NK4 * 1024);
    if (s > 0) {
#pragma unroll
      for (int g = 0; g < 4; ++g) lstm_local_publish(gb, g * H + u0, sq, sg[g][0], sg[g][1], sg[g][2], sg[g][3]);
    }
    if (rvalid) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t o = o0 + (uint32_t)r * uP;
        dG[o] = sg[0][r]; dG[o + HP] = sg[1][r]; dG[o + 2 * HP] = sg[2][r]; dG[o + 3 * HP] = sg[3][r];
      }
    }
    if (s > 0) {
      lstm_lds_barrier();
      lstm_f32x4 p[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) p[q] = lstm_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NK4; ++ks) {                      // four independent chains
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(gb + ks * 1024 + lane * 16);
        p[ks & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wb[ks]), bh, p[ks & 3], 0, 0, 0);
      }
      // the next step's operands (issuing them in front of the barrier, under the MFMA chain, measured slower: 683 -> 814 us at 8 sequences)
      const int s1 = s - 1, s2 = s1 > 0 ? s1 - 1 : 0;
      const uint32_t on = ubase + (uint32_t)((dir == 0 ? s1 : T - 1 - s1) * Bn);
      const uint32_t opn = ubase + (uint32_t)((dir == 0 ? s2 : T - 1 - s2) * Bn);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t du = (uint32_t)r * uP;
        sg[0][r] = gates[on + du]; sg[1][r] = gates[on + du + HP]; sg[2][r] = gates[on + du + 2 * HP]; sg[3][r] = gates[on + du + 3 * HP];
        cp[r] = cst[opn + du];
        gy[r] = goutp[on + du];
      }
      rec = (p[0] + p[1]) + (p[2] + p[3]);
    }
  }
}

static int64_t lstm_cluster_uint4_per_dir(int H) {
  const int64_t nf = (int64_t)(H / 32) * 4 * (H / 16) * 64, nb = (int64_t)(H / 32) * (4 * H / 16) * 64;
  return 2 * nf + 2 * nb;
}
// a direction's pack: the cluster-form fragments (hi, lo), then the single-workgroup form's (H <= 192)
static int64_t lstm_pack_uint4_per_dir(int H) { return lstm_cluster_uint4_per_dir(H) + lstm_local_uint4_per_dir(H); }
// launches of the single-workgroup form: one workgroup per (16-sequence tile, direction)
// Which launches take the single-workgroup form (H = 192, T = 256, kernel time alone): forward 513 / 519 / 540 / 579 us at 8 / 16 /
// 32 / 64 sequences against 656 - 681 us for the cluster form; backward 683 - 778 / 870 / 1150 / 1207 - 1350 us against 896 - 982.  Its
// 40 memory instructions per wave and step (64-byte runs: sixteen sequences of one tensor row) queue up as more workgroups run, so:
// forward up to RFX_LSTM_LOCAL sequences (default 64), backward up to RFX_LSTM_LOCAL_BWD (default 16).  In the Demucs step the
// layer-4 BLSTM sees 3 x the clips (torchaudio _BLSTM unfolds 256 frames into three overlapping 200-frame chunks): 64 clips stay on
// the cluster form (the single-workgroup form measured +0.5 ms there), 8 clips per rank take the forward form (-0.8 ms of 33).
static int lstm_local_m[2] = {-1, -1};
static int lstm_local_max_bn(bool bwd) {
  int* m = lstm_local_m;
  if (m[0] < 0) {
    const char* e = getenv("RFX_LSTM_LOCAL");
    const char* eb = getenv("RFX_LSTM_LOCAL_BWD");
    m[0] = e ? atoi(e) : 64;
    m[1] = m[0] > 0 ? (eb ? atoi(eb) : 16) : 0;
  }
  return m[bwd];
}
// Pin the form choice from the host (tests compare a batch with its single clips on ONE form: the two forms round h_t to bf16 after
// differently ordered fp32 sums, which a 200-step recurrence amplifies to bf16-level differences).  Negative: back to the
// environment / defaults.
extern "C" int rfx_lstm_set_local(int32_t fwd_max_sequences, int32_t bwd_max_sequences) {
  if (fwd_max_sequences < 0 || bwd_max_sequences < 0) { lstm_local_m[0] = lstm_local_m[1] = -1; return 0; }
  lstm_local_m[0] = fwd_max_sequences;
  lstm_local_m[1] = fwd_max_sequences > 0 ? bwd_max_sequences : 0;
  return 0;
}
static bool lstm_local_ok(int H, int prec, int Bn, bool bwd) { return prec == RFX_PREC_BF16 && H == 192 && Bn <= lstm_local_max_bn(bwd); }
template <int NW>
static int lstm_local_launch(bool bwd, LstmArgs a, hipStream_t s) {
  constexpr int H = 16 * NW;
  const int ntiles = (a.Bn + 15) / 16;
  const uint4* lpack = a.packA + lstm_cluster_uint4_per_dir(H);       // behind the direction's cluster-form fragments
  const int64_t lstride = a.pack_stride;
  const size_t smem = (size_t)2 * (bwd ? 4 : 1) * (H / 32) * 1024;
  static bool attr[2] = {false, false};
  if (!attr[bwd]) {
    const void* k = bwd ? reinterpret_cast<const void*>(&lstm_bwd_local_kernel<NW>) : reinterpret_cast<const void*>(&lstm_fwd_local_kernel<NW>);
    if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return -3;
    attr[bwd] = true;
  }
  if (bwd) hipLaunchKernelGGL((lstm_bwd_local_kernel<NW>), dim3(2 * ntiles), dim3(64 * NW), smem, s, a, lpack, lstride);
  else hipLaunchKernelGGL((lstm_fwd_local_kernel<NW>), dim3(2 * ntiles), dim3(64 * NW), smem, s, a, lpack, lstride);
  RFX_CHECK_LAUNCH();
  return 0;
}

static bool lstm_bad_h(int H) { return H <= 0 || H % 32 || H > 512; }
static size_t lstm_smem_bytes(int H) { return (size_t)(H / 32) * 4096; }   // fwd: 2*nks KB, bwd: 4*nwc KB (equal)
// workspace: [0,256) error flag | counters (64 B per cluster) | exchange buffers sized for the backward sweep
static int64_t lstm_ws_ctr_off() { return 256; }
static int64_t lstm_ws_xch_off(int H) { return 256 + (int64_t)(RFX_LSTM_MAX_WAVES / (H / 32)) * 64; }

extern "C" int rfx_lstm_pack_bytes(int32_t H) {
  if (lstm_bad_h(H)) return -1;
  return (int)(lstm_pack_uint4_per_dir(H) * 16);
}

extern "C" int rfx_lstm_ws_bytes(int32_t H) {
  if (lstm_bad_h(H)) return -1;
  const int64_t nwc = H / 32;
  return (int)(lstm_ws_xch_off(H) + (RFX_LSTM_MAX_WAVES / nwc) * 2 * nwc * nwc * 4096);
}

// Host glue of one bidirectional layer as single launches (they were 2 cat + 2 add launches per forward call and 6 add_ launches per
// backward call: 64 of the Demucs step's launches).
//   cat:     wcat [8H][Cin] = [w_ih ; w_ih_r],  bcat [8H] = [b_ih + b_hh ; b_ih_r + b_hh_r]  (the operands of the input-projection GEMM)
//   scatter: the projection GEMM's weight / bias gradients added into the six parameter gradients they belong to (both biases of a
//            direction receive the same gradient)
__global__ __launch_bounds__(256) void lstm_cat_kernel(const float* __restrict__ w0, const float* __restrict__ w1, const float* __restrict__ bi0,
                                                       const float* __restrict__ bh0, const float* __restrict__ bi1, const float* __restrict__ bh1,
                                                       int64_t nw, int nb, float* __restrict__ wcat, float* __restrict__ bcat) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < 2 * nw; i += stride) wcat[i] = i < nw ? w0[i] : w1[i - nw];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < 2 * nb; i += stride)
    bcat[i] = i < nb ? bi0[i] + bh0[i] : bi1[i - nb] + bh1[i - nb];
}
__global__ __launch_bounds__(256) void lstm_scatter_kernel(const float* __restrict__ dwcat, const float* __restrict__ dbcat, int64_t nw, int nb,
                                                           float* __restrict__ gw0, float* __restrict__ gw1, float* __restrict__ gbi0,
                                                           float* __restrict__ gbh0, float* __restrict__ gbi1, float* __restrict__ gbh1) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < 2 * nw; i += stride) {
    if (i < nw) gw0[i] += dwcat[i]; else gw1[i - nw] += dwcat[i];
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < 2 * nb; i += stride) {
    const float v = dbcat[i];
    if (i < nb) { gbi0[i] += v; gbh0[i] += v; } else { gbi1[i - nb] += v; gbh1[i - nb] += v; }
  }
}
extern "C" int rfx_lstm_cat_params(const float* w_ih, const float* w_ih_r, const float* b_ih, const float* b_hh, const float* b_ih_r,
                                   const float* b_hh_r, int32_t H, int32_t Cin, float* wcat, float* bcat, void* stream) {
  if (!w_ih || !w_ih_r || !b_ih || !b_hh || !b_ih_r || !b_hh_r || !wcat || !bcat || H <= 0 || Cin <= 0) return -1;
  const int64_t nw = (int64_t)4 * H * Cin;
  const int grid = (int)((2 * nw + 255) / 256 < 1024 ? (2 * nw + 255) / 256 : 1024);
  hipLaunchKernelGGL(lstm_cat_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w_ih, w_ih_r, b_ih, b_hh, b_ih_r, b_hh_r, nw, 4 * H, wcat, bcat);
  RFX_CHECK_LAUNCH();
  return 0;
}
extern "C" int rfx_lstm_grad_scatter(const float* dwcat, const float* dbcat, int32_t H, int32_t Cin, float* g_w_ih, float* g_w_ih_r,
                                     float* g_b_ih, float* g_b_hh, float* g_b_ih_r, float* g_b_hh_r, void* stream) {
  if (!dwcat || !dbcat || !g_w_ih || !g_w_ih_r || !g_b_ih || !g_b_hh || !g_b_ih_r || !g_b_hh_r || H <= 0 || Cin <= 0) return -1;
  const int64_t nw = (int64_t)4 * H * Cin;
  const int grid = (int)((2 * nw + 255) / 256 < 1024 ? (2 * nw + 255) / 256 : 1024);
  hipLaunchKernelGGL(lstm_scatter_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dwcat, dbcat, nw, 4 * H, g_w_ih, g_w_ih_r, g_b_ih, g_b_hh,
                     g_b_ih_r, g_b_hh_r);
  RFX_CHECK_LAUNCH();
  return 0;
}

extern "C" int rfx_lstm_pack(const float* whh, int32_t H, void* pack, void* stream) {
  if (!whh || !pack || lstm_bad_h(H)) return -1;
  const int64_t nf = (int64_t)(H / 32) * 4 * (H / 16) * 64;
  uint4* p = reinterpret_cast<uint4*>(pack);
  const int64_t total = lstm_cluster_uint4_per_dir(H) / 2;
  const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(lstm_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, whh, H, p, p + 2 * nf);
  RFX_CHECK_LAUNCH();
  return 0;
}

// 1 if rfx_lstm_fwd (bwd = 0) / rfx_lstm_bwd (bwd = 1) will take the single-workgroup form for this shape: the caller must then have
// packed its fragments too (rfx_lstm_pack_local; the same `pack` buffer, behind the cluster form's)
extern "C" int rfx_lstm_local(int32_t H, int32_t Bn, int32_t prec, int32_t bwd) {
  if (lstm_bad_h(H) || Bn <= 0) return 0;
  return lstm_local_ok(H, prec, Bn, bwd != 0) ? 1 : 0;
}

extern "C" int rfx_lstm_pack_local(const float* whh, int32_t H, void* pack, void* stream) {
  if (!whh || !pack || lstm_bad_h(H)) return -1;
  const int64_t nl = lstm_local_uint4_per_dir(H);
  if (nl == 0) return 0;
  uint4* p = reinterpret_cast<uint4*>(pack);
  const int gl = (int)((nl + 255) / 256 < 2048 ? (nl + 255) / 256 : 2048);
  hipLaunchKernelGGL(lstm_pack_local_kernel, dim3(gl), dim3(256), 0, (hipStream_t)stream, whh, H, p + lstm_cluster_uint4_per_dir(H));
  RFX_CHECK_LAUNCH();
  return 0;
}

// co-residency bound: every wave of a launch must be on the machine at once (they wait for each other).  The bound is taken for
// THE instantiation being launched (the register-resident forms use more VGPRs than the generic ones); a launch that cannot be
// co-resident is refused, not attempted.  wpc = waves per cluster (forward: H/16, backward: H/32).
template <typename K>
static int lstm_launch(K kernel, LstmArgs a, int wpc, void* ws, void* stream) {
  const int H = a.H, ntiles = (a.Bn + 31) / 32;
  int mc = 0;
  {
    static thread_local const void* seen[24];
    static thread_local int seen_mc[24], seen_h[24];
    static thread_local int nseen = 0;
    int i = 0;
    for (; i < nseen; ++i) if (seen[i] == reinterpret_cast<const void*>(kernel) && seen_h[i] == H) break;
    if (i == nseen) {
      if (nseen == 24) nseen = 0;
      int dev = 0, ncu = 0, occ = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
          hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, 64, lstm_smem_bytes(H)) != hipSuccess) return -3;
      int waves = ncu * occ;
      if (waves > RFX_LSTM_MAX_WAVES) waves = RFX_LSTM_MAX_WAVES;
      i = nseen++;
      seen[i] = reinterpret_cast<const void*>(kernel);
      seen_h[i] = H;
      seen_mc[i] = (waves / wpc) & ~7;
    }
    mc = seen_mc[i];
  }
  const int mc_ws = RFX_LSTM_MAX_WAVES / (H / 32);       // clusters the workspace (counters, exchange buffers) is laid out for
  if (mc > mc_ws) mc = mc_ws & ~7;
  if (mc < 8) return -4;
  const size_t smem = lstm_smem_bytes(H);
  unsigned char* w = reinterpret_cast<unsigned char*>(ws);
  a.err = reinterpret_cast<int32_t*>(w);
  a.ctr = reinterpret_cast<uint32_t*>(w + lstm_ws_ctr_off());
  a.xch = reinterpret_cast<uint64_t*>(w + lstm_ws_xch_off(H));
  hipStream_t s = (hipStream_t)stream;
  for (int t0 = 0; t0 < ntiles; t0 += mc / 2) {
    const int nt = ntiles - t0 < mc / 2 ? ntiles - t0 : mc / 2;
    a.tile0 = t0;
    a.nclusters = 2 * nt;
    if (hipMemsetAsync(a.ctr, 0, (size_t)mc * 64, s) != hipSuccess) return -3;
    const int blocks = ((a.nclusters + 7) / 8) * 8 * wpc;
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(64), smem, s, a);
    RFX_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int rfx_lstm_fwd(const float* xp, const void* pack, int32_t T, int32_t Bn, int32_t H, float* out,
                            float* gates, float* cstate, void* ws, int32_t prec, void* stream) {
  if (!xp || !pack || !out || !ws || T <= 0 || Bn <= 0 || lstm_bad_h(H)) return -1;
  if ((gates == nullptr) != (cstate == nullptr)) return -1;
  if ((int64_t)4 * H * T * Bn >= ((int64_t)1 << 31)) return -1;
  LstmArgs a{};
  a.xp = xp; a.packA = reinterpret_cast<const uint4*>(pack); a.out = out; a.gates = gates; a.cstate = cstate;
  a.pack_stride = lstm_pack_uint4_per_dir(H); a.T = T; a.Bn = Bn; a.H = H; a.P = T * Bn;
  const int wf = H / 16;
  if (lstm_local_ok(H, prec, Bn, false)) return lstm_local_launch<12>(false, a, (hipStream_t)stream);
  if (prec == RFX_PREC_BF16) {                 // bf16 operands (bf16-mixed): no lo fragments
    // the wave's whole W_hh slice (2 tiles x H/16 k-steps: 96 VGPRs at H = 192, 192 at H = 384) stays in registers instead of being
    // re-streamed through L1 every time step
    if (H == 192) return lstm_launch(lstm_fwd_kernel<12, 12, false>, a, wf, ws, stream);
    if (H == 256) return lstm_launch(lstm_fwd_kernel<16, 16, false>, a, wf, ws, stream);
    if (H == 384) return lstm_launch(lstm_fwd_kernel<24, 24, false>, a, wf, ws, stream);
    return H % 64 == 0 ? lstm_launch(lstm_fwd_kernel<4, 0, false>, a, wf, ws, stream)
                       : lstm_launch(lstm_fwd_kernel<2, 0, false>, a, wf, ws, stream);
  }
  if (H == 192) return lstm_launch(lstm_fwd_kernel<12, 12, true>, a, wf, ws, stream);    // hi + lo fragments: 192 VGPRs
  if (H == 256) return lstm_launch(lstm_fwd_kernel<4, 16, true>, a, wf, ws, stream);     // HDemucs DConv widths and Open-Unmix: unrolled k loop
  if (H == 384) return lstm_launch(lstm_fwd_kernel<4, 24, true>, a, wf, ws, stream);
  return H % 64 == 0 ? lstm_launch(lstm_fwd_kernel<4, 0, true>, a, wf, ws, stream) : lstm_launch(lstm_fwd_kernel<2, 0, true>, a, wf, ws, stream);
}

extern "C" int rfx_lstm_bwd(const float* gout, const void* pack, const float* gates, const float* cstate, int32_t T,
                            int32_t Bn, int32_t H, float* dG, void* ws, int32_t prec, void* stream) {
  if (!gout || !pack || !gates || !cstate || !dG || !ws || T <= 0 || Bn <= 0 || lstm_bad_h(H)) return -1;
  if ((int64_t)4 * H * T * Bn >= ((int64_t)1 << 31)) return -1;
  LstmArgs a{};
  a.gout = gout; a.packA = reinterpret_cast<const uint4*>(pack); a.gates = const_cast<float*>(gates);
  a.cstate = const_cast<float*>(cstate); a.dG = dG;
  a.pack_stride = lstm_pack_uint4_per_dir(H); a.T = T; a.Bn = Bn; a.H = H; a.P = T * Bn;
  const int wb = H / 16;
  if (lstm_local_ok(H, prec, Bn, true)) return lstm_local_launch<12>(true, a, (hipStream_t)stream);
  if (prec == RFX_PREC_BF16) {
    // single-fragment products: the wave's W_hh^T slice (4 fragments per output tile: 96 VGPRs at H = 192, 192 at H = 384) is resident
    if (H == 192) return lstm_launch(lstm_bwd_kernel<3, 6, false, true>, a, wb, ws, stream);
    if (H == 256) return lstm_launch(lstm_bwd_kernel<4, 8, false, true>, a, wb, ws, stream);
    if (H == 384) return lstm_launch(lstm_bwd_kernel<4, 12, false, true>, a, wb, ws, stream);
    return (H / 32) % 2 == 0 ? lstm_launch(lstm_bwd_kernel<2, 0, false, false>, a, wb, ws, stream)
                             : lstm_launch(lstm_bwd_kernel<1, 0, false, false>, a, wb, ws, stream);
  }
  if (H == 192) return lstm_launch(lstm_bwd_kernel<3, 6, true, true>, a, wb, ws, stream);      // hi + lo resident: 192 VGPRs
  return (H / 32) % 2 == 0 ? lstm_launch(lstm_bwd_kernel<2, 0, true, false>, a, wb, ws, stream)
                           : lstm_launch(lstm_bwd_kernel<1, 0, true, false>, a, wb, ws, stream);
}
