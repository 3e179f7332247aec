// Framed real FFT kernels: STFT / iSTFT forward and backward for every geometry
// of the RemFX hot path (n_fft 512..4096, arbitrary hop, hann window of `win`
// samples centred in n_fft, centre + reflect padding).
//
// Replaces torch.stft / torch.istft call sites: utils.py:148-154 (spectrogram),
// HDemucs _spec/_ispec (models.py:319), auraloss STFTLoss (models.py:320 ...),
// Open-Unmix Separator (models.py:298), MelSpectrogram (classifier.py:200).
//
// Structure (round 3: register passes; rounds 1-2 ran six radix-4 passes in place in LDS and were VALU-issue-bound on their index
// arithmetic, ~200 instructions per complex point):
//   * a real n_fft-point transform is an NC = n_fft/2 point complex FFT of z[n] = x[2n] + i x[2n+1] plus the split / merge step;
//   * one workgroup = 256 threads = FB = 4096 / NC frames at a time, T = NC / 16 threads per frame, every thread owns 16 complex
//     points of its frame in REGISTERS in each pass.  NC = RA * 16 * 16: pass A (radix RA = NC / 256 = 2, 4, 8, stride 256; absent for
//     NC = 256) takes its input straight from global memory, passes B and C are radix-16 butterflies (two radix-4 stages), three
//     LDS exchanges in all (A->B, B->C, C->split) instead of seven read-modify-write passes.  Complex values are 2-vectors, so the
//     butterflies compile to packed fp32 instructions (v_pk_add_f32 / v_pk_fma_f32: a complex add is one instruction, a complex
//     multiply two);
//   * every exchange has its own LDS layout (E1, E2, E3 below), chosen so that the writing pass and the reading pass are both
//     lane-contiguous / at most 2-way bank-conflicted, and every LDS access is base + immediate offset (scripts/probes/fft_model.py
//     is the index model of this file, checked against numpy.fft);
//   * twiddles: W_256^(j m) of pass B and the 16 window pairs of a thread are frame-independent and stay in registers while the
//     workgroup walks `nbatch` consecutive frame batches; pass A chains its RA - 1 twiddles from one table value per butterfly.
//     The tables (host-computed in double precision, one set per n_fft and device) are read from global memory once per workgroup;
//   * the inverse transform is the same forward code on conjugated data (the merge step writes conj Z, the overlap-add reads conj z);
//   * the transposing epilogue (lanes along frames -> [bin][frame] stores) works on PAIRS of bins (k, NC - k), which share their
//     two LDS reads; workgroups that write the same (row, bin) lines are placed on the same XCD (block b runs on XCD b % 8) so
//     partial-line stores merge in one L2.
#include "common.h"
#include <algorithm>
#include <climits>
#include <math.h>
#include <stdlib.h>
#ifndef RFX_FFT_PAIR_OCC
// waves per SIMD the pair-loss kernel is compiled for where three workgroups fit a CU's LDS (n_fft 2048 / 4096; n_fft 1024 needs
// 53 KB): 3 = 168 registers + 44 spilled instead of 220, MRSTFT forward + backward at 64 clips 7.80 -> 7.23 ms (A/B:
// scripts/build_abl.py fft RFX_FFT_PAIR_OCC 2, scripts/perf_loss.py)
#define RFX_FFT_PAIR_OCC 3
#endif
#include <mutex>

typedef float v2f __attribute__((ext_vector_type(2)));

struct FftArgs {
  rfx_stft_desc d;
  const float* x;       // analysis: signal [R][T];   synthesis: spectrum
  const float* window;  // [win]
  const float* mul;     // optional per padded-sample multiplier (iSTFT 1/envelope)
  float* out;           // analysis: spectrum;       synthesis: signal [R][T] (atomic accumulate)
  const v2f* tables;    // [256] W_256^t | [256] W_NC^t | [NC/2 + 1] e^{-i pi k / NC}
  int groups_per_row;   // workgroups per row
  int nbatch;           // frame batches (FB frames each) a workgroup walks
  uint32_t hop_magic;   // ceil(2^32 / hop): q / hop = umulhi(q, hop_magic) for the q < 2^20 of one batch's span (hop <= 4096; else 0)
  // synthesis, RFX_STFT_COMPLEX_FM only (rfx_fft_synthesis_lossgrad): x = the PREDICTION's spectrum, and what is transformed is the
  // STFTLoss gradient at it -- the value rfx_stft_loss_grad_m would have written, computed where the merge step loads it
  const float* lg_ymag;   // clamped target magnitudes [R][frames][bins]; NULL = x is the spectrum to transform
  const float* lg_sums;   // [R][3] row sums of the forward
  const float* lg_gup;    // optional device scalar multiplying both weights
  float lg_wsc, lg_wlm, lg_eps;
  // synthesis by OWNERSHIP (round 6): a workgroup walks `walk_frames` consecutive frames of a row (the last walk of a row takes the
  // remainder) after `halo_frames` frames of run-in, and stores every output sample of its range exactly once
  int walk_frames, halo_frames;
  float* ws;              // OWN scratch: zone[R][2][n_fft + 1] | carry[grid][2][n_fft]
  int64_t cover_lo, cover_hi;   // padded positions some frame covers
};

// d [ w_sc sqrt(A) / sqrt(B) + w_lm sum |log|X| - log|Y|| ] / dX at one cell: stft_loss_grad_kernel's formula (csrc/losses.hip, paired
// form) with its three divisions by |X| replaced by one reciprocal (1-ulp differences; |X| == |Y| still gives exactly 0: the
// magnitude comes out of the same rfx_pow2 + hardware sqrt sequence as the forward's)
__device__ __forceinline__ v2f fft_lossgrad_cell(v2f x, float ym, float ksc, float wlm, float eps) {
  const float px = rfx_pow2(x.x, x.y);
  const float xm = __builtin_amdgcn_sqrtf(fmaxf(px, eps));
  const float r = 1.0f / xm;
  const float sg = xm > ym ? wlm : (xm < ym ? -wlm : 0.f);
  float sc = fmaf(ksc, xm - ym, sg * r) * r;
  sc = px > eps ? sc : 0.f;                                   // clamp(min = eps) passes no gradient below eps
  return v2f{sc * x.x, sc * x.y};
}

template <int LOGN>
struct FftCfg {
  static constexpr int NC = 1 << LOGN, T = NC / 16, FB = 256 / T, RA = NC / 256, NB = RA > 1 ? 16 / RA : 1;
  static constexpr int FS = NC + (RA > 1 ? 256 : 32) + 2;        // frame stride (elements); = 2 mod 32: frames interleave in the epilogue
  static constexpr int RS1 = T + 16, RS2 = T + 1, RS3 = 256 + (RA > 1 ? 32 / RA : 0);
  static constexpr int NH = NC / 2 + 1;
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding GLOBAL access (vmcnt(0)): the
// stores of a batch's epilogue and the prefetched loads of the next batch would be drained at each of the ~7 barriers of a batch.
// Every barrier in this file separates LDS writes from LDS reads; global data is never exchanged between threads.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ v2f cmul(v2f a, v2f b) { return a.xx * b + a.yy * v2f{-b.y, b.x}; }
__device__ __forceinline__ v2f mul_negi(v2f a) { return v2f{a.y, -a.x}; }
__device__ __forceinline__ v2f cconj(v2f a) { return v2f{a.x, -a.y}; }

// forward DFTs over x[0], x[S], ..., x[(R-1) S] in place; output Y[m] lands at x[fft_slot<R>(m) * S]
template <int R> __device__ __forceinline__ constexpr int fft_slot(int m) {
  return R == 16 ? (m & 3) * 4 + (m >> 2) : R == 8 ? (m & 3) * 2 + (m >> 2) : m;
}
template <int S> __device__ __forceinline__ void dft2(v2f* x) {
  const v2f t = x[0] - x[S];
  x[0] = x[0] + x[S];
  x[S] = t;
}
template <int S> __device__ __forceinline__ void dft4(v2f* x) {
  const v2f t0 = x[0] + x[2 * S], t1 = x[0] - x[2 * S], t2 = x[S] + x[3 * S], t3 = mul_negi(x[S] - x[3 * S]);
  x[0] = t0 + t2;
  x[S] = t1 + t3;
  x[2 * S] = t0 - t2;
  x[3 * S] = t1 - t3;
}
template <int R, int S> __device__ __forceinline__ void dftR(v2f* x) {
  constexpr float H = 0.70710678118654752440f, C1 = 0.92387953251128675613f, S1 = 0.38268343236508977173f;
  if (R == 2) dft2<S>(x);
  else if (R == 4) dft4<S>(x);
  else if (R == 8) {                       // q = q1 + 2 q2:  radix-4 over q2, W8^(q1 m2), radix-2 over q1
    dft4<2 * S>(x);
    dft4<2 * S>(x + S);
    x[3 * S] = cmul(x[3 * S], v2f{H, -H});
    x[5 * S] = mul_negi(x[5 * S]);
    x[7 * S] = cmul(x[7 * S], v2f{-H, -H});
#pragma unroll
    for (int m2 = 0; m2 < 4; ++m2) dft2<S>(x + 2 * m2 * S);
  } else {                                 // q = q1 + 4 q2:  radix-4 over q2, W16^(q1 m2), radix-4 over q1
#pragma unroll
    for (int q1 = 0; q1 < 4; ++q1) dft4<4 * S>(x + q1 * S);
    x[5 * S] = cmul(x[5 * S], v2f{C1, -S1});
    x[9 * S] = cmul(x[9 * S], v2f{H, -H});
    x[13 * S] = cmul(x[13 * S], v2f{S1, -C1});
    x[6 * S] = cmul(x[6 * S], v2f{H, -H});
    x[10 * S] = mul_negi(x[10 * S]);
    x[14 * S] = cmul(x[14 * S], v2f{-H, -H});
    x[7 * S] = cmul(x[7 * S], v2f{S1, -C1});
    x[11 * S] = cmul(x[11 * S], v2f{-H, -H});
    x[15 * S] = cmul(x[15 * S], v2f{-C1, S1});
#pragma unroll
    for (int m2 = 0; m2 < 4; ++m2) dft4<S>(x + 4 * m2 * S);
  }
}

// padded-signal coordinate p = f*hop + t  ->  index into x (or -1)
__device__ __forceinline__ int map_sample(const rfx_stft_desc& d, int p) {
  if (d.in_mode == 0) {  // centre + reflect (on top of an optional extra reflect pad)
    const int Tp = d.T + d.extra_pad_l + d.extra_pad_r;
    int s = p - d.n_fft / 2;
    if (s < 0) s = -s;
    if (s >= Tp) s = 2 * (Tp - 1) - s;
    s -= d.extra_pad_l;
    if (s < 0) s = -s;
    if (s >= d.T) s = 2 * (d.T - 1) - s;
    return (s >= 0 && s < d.T) ? s : -1;
  }
  const int s = p - d.in_offset;  // iSTFT: centre trim + crop folded into one offset
  return (s >= 0 && s < d.T) ? s : -1;
}

template <int LOGN>
__device__ __forceinline__ void block_coords(const FftArgs& a, int& row, int& f_first) {
  typedef FftCfg<LOGN> K;
  // same-row frame groups on the same XCD (block b -> XCD b % 8), adjacent in time
  const int b = blockIdx.x, xcd = b & 7, q = b >> 3;
  const int slot = q / a.groups_per_row, grp = q - slot * a.groups_per_row;
  row = slot * 8 + xcd;
  f_first = a.d.frame0 + grp * (K::FB * a.nbatch);
}

// Twiddles of the passes.  Pass B: W_256^(j2 m) from an LDS table stored [m][j2], so that a thread's 15 reads are base + immediate
// offset (registers would cost 30 VGPRs per thread: the kernel is register-bound at three waves per SIMD).  Pass A: the base
// W_NC^(j1) of each butterfly, from LDS for RA >= 4 (2 / 4 butterflies); for RA = 2 (LDS is full at NC = 512) W_512^u times a constant.
template <int LOGN>
struct FftState {
  const v2f* twB;                                        // LDS, + (u & 15)
  const v2f* twAs;                                       // LDS, + u (RA >= 4)
  v2f twA0;                                              // RA = 2: W_512^u; butterfly h uses W_512^(u + 32 h) = twA0 W_16^h
};
template <int LOGN>
struct FftTables {
  v2f twH[FftCfg<LOGN>::NH];
  v2f tw256t[256];
  v2f twAs[FftCfg<LOGN>::RA >= 4 ? 256 : 1];
};
template <int LOGN>
__device__ __forceinline__ void fft_setup(const FftArgs& a, FftTables<LOGN>& tb, FftState<LOGN>& st) {
  typedef FftCfg<LOGN> K;
  const int tid = threadIdx.x, u = tid & (K::T - 1);
  {                                                        // all loads first: a load -> wait -> LDS write loop costs a round trip each
    constexpr int NI = (K::NH + 255) / 256;
    v2f tmp[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) tmp[i] = a.tables[512 + min(tid + 256 * i, K::NH - 1)];
#pragma unroll
    for (int i = 0; i < NI; ++i) if (tid + 256 * i < K::NH) tb.twH[tid + 256 * i] = tmp[i];
  }
  tb.tw256t[tid] = a.tables[(tid & 15) * (tid >> 4)];    // [m][j2]
  if (K::RA >= 4) tb.twAs[tid] = a.tables[256 + tid];
  st.twB = tb.tw256t + (u & 15);
  st.twAs = tb.twAs + u;
  st.twA0 = a.tables[256 + (K::RA == 2 ? u : 0)];
}

// The three passes on one batch: z = the thread's 16 points i = u + T n (n = 0..15) of frame fl.  Result: Z[k] in layout E3.
// Callers put a barrier before the first LDS read of the NEXT use of `data`.
template <int LOGN>
__device__ __forceinline__ void fft_core(v2f (&z)[16], v2f* data, const FftState<LOGN>& st, const v2f* wout = nullptr) {
  typedef FftCfg<LOGN> K;
  const int tid = threadIdx.x, u = tid & (K::T - 1), fl = tid / K::T;
  v2f* fr = data + fl * K::FS;
  if (K::RA > 1) {
    // pass A: butterfly h works on z[h + NB q], j1 = u + h T;  E1: element (b1 = m, j1) at (j1 / 16) RS1 + 16 m + j1 % 16
    v2f* wA = fr + (u >> 4) * K::RS1 + (u & 15);
#pragma unroll
    for (int h = 0; h < K::NB; ++h) {
      dftR<K::RA, K::NB>(z + h);
      v2f w[K::RA > 1 ? K::RA : 2];
      if (K::RA == 2) {
        constexpr float CS[8][2] = {{1.f, 0.f}, {0.92387953251128675613f, -0.38268343236508977173f},
                                    {0.70710678118654752440f, -0.70710678118654752440f}, {0.38268343236508977173f, -0.92387953251128675613f},
                                    {0.f, -1.f}, {-0.38268343236508977173f, -0.92387953251128675613f},
                                    {-0.70710678118654752440f, -0.70710678118654752440f}, {-0.92387953251128675613f, -0.38268343236508977173f}};
        v2f b = st.twA0;
        asm volatile("" : "+v"(b));                      // keeps the eight products out of loop-invariant registers
        w[1] = cmul(b, v2f{CS[h & 7][0], CS[h & 7][1]});
      } else w[1] = st.twAs[h * K::T];
#pragma unroll
      for (int m = 2; m < K::RA; ++m) w[m] = (m & 1) ? cmul(w[m - 1], w[1]) : cmul(w[m / 2], w[m / 2]);
      wA[h * (K::T / 16) * K::RS1] = z[h + K::NB * fft_slot<K::RA>(0)];
#pragma unroll
      for (int m = 1; m < K::RA; ++m)
        wA[h * (K::T / 16) * K::RS1 + 16 * m] = cmul(z[h + K::NB * fft_slot<K::RA>(m)], w[m]);
    }
    lds_barrier();
#pragma unroll
    for (int q = 0; q < 16; ++q) z[q] = fr[q * K::RS1 + u];
    lds_barrier();
  }
  // pass B: radix 16 over stride 16 of a 256-point block b1 = u / 16, j2 = u % 16;  E2: element (v3 = 16 b1 + m, q = j2) at q RS2 + v3
  dftR<16, 1>(z);
  {
    v2f* wB = fr + (u & 15) * K::RS2 + (u >> 4) * 16;
    wB[0] = z[fft_slot<16>(0)];
#pragma unroll
    for (int m = 1; m < 16; ++m) wB[m] = cmul(z[fft_slot<16>(m)], st.twB[16 * m]);
  }
  lds_barrier();
#pragma unroll
  for (int q = 0; q < 16; ++q) z[q] = fr[q * K::RS2 + u];
  lds_barrier();
  // pass C: radix 16, no twiddle;  output k = mA + RA (mB + 16 m), mA = u / 16, mB = u % 16;  E3: (k % RA) RS3 + k / RA
  dftR<16, 1>(z);
  {
    v2f* wC = fr + (u >> 4) * K::RS3 + (u & 15);
#pragma unroll
    for (int m = 0; m < 16; ++m) wC[16 * m] = wout ? z[fft_slot<16>(m)] * wout[m] : z[fft_slot<16>(m)];   // wout: a register array or null (inlined)
  }
}
template <int LOGN> __device__ __forceinline__ int fft_phys3(int k) {
  typedef FftCfg<LOGN> K;
  return K::RA > 1 ? (k & (K::RA - 1)) * K::RS3 + (k >> (LOGN - 8)) : k;
}

template <int LOGN>
__global__ __launch_bounds__(256, 3) void fft_analysis_kernel(const FftArgs a) {
  typedef FftCfg<LOGN> K;
  constexpr int NC = K::NC, T = K::T, FB = K::FB, N = 2 * NC;
  __shared__ v2f data[FB * K::FS];
  __shared__ FftTables<LOGN> tb;
  const rfx_stft_desc& d = a.d;
  int row, f_first;
  block_coords<LOGN>(a, row, f_first);
  if (row >= d.R) return;
  const int tid = threadIdx.x, u = tid & (T - 1), fl = tid / T;
  FftState<LOGN> st;
  fft_setup<LOGN>(a, tb, st);
  lds_barrier();                       // the passes read the tables other threads loaded
  const int f_end = d.frame0 + d.frames_out;
  const float* xr = a.x + (int64_t)row * d.T;
  const int woff = (N - d.win) / 2;
  // the thread's 16 window pairs (x scale; 0 outside the window): frame-independent
  v2f wv[16];
  {   // 32 independent loads at clamped indices, validity applied to the VALUE (a load inside `ok ? w[i] : 0` becomes a branch
      // with its own s_waitcnt: 32 serialised round trips = ~10 us per workgroup, measured as an "empty" kernel of 100-150 us)
    float wl[16][2];
#pragma unroll
    for (int n = 0; n < 16; ++n)
#pragma unroll
      for (int c = 0; c < 2; ++c) wl[n][c] = a.window[min(max(2 * (u + T * n) + c - woff, 0), d.win - 1)];
#pragma unroll
    for (int n = 0; n < 16; ++n) {
      const int w0 = 2 * (u + T * n) - woff;
      wv[n] = v2f{((w0 >= 0) & (w0 < d.win)) ? wl[n][0] * d.scale : 0.f, ((w0 + 1 >= 0) & (w0 + 1 < d.win)) ? wl[n][1] * d.scale : 0.f};
    }
  }
  const int shift = d.in_mode == 0 ? d.n_fft / 2 + d.extra_pad_l : d.in_offset;
  const int FO = d.frames_out;
  // workgroup-uniform: every frame of a batch lies inside the signal (no reflection, no frame past the end) and its sample
  // pairs are 8-byte aligned -> plain loads straight into the registers of pass A, ISSUED BEFORE the previous batch's epilogue
  // (the barriers do not wait for them).  Otherwise (the first / last batches of a row, odd hops) the samples are mapped one by one
  // in a rolled loop and staged through LDS.
  auto batch_fast = [&](int fb0) -> bool {
    const int64_t pb0 = (int64_t)fb0 * d.hop;
    return (fb0 + FB <= f_end) & (pb0 - shift >= 0) & (pb0 + (int64_t)(FB - 1) * d.hop + N - 1 - shift < (int64_t)d.T) &
           !(d.hop & 1) & ((((int64_t)row * d.T + pb0 - shift) & 1) == 0) & ((reinterpret_cast<uintptr_t>(a.x) & 7) == 0);
  };
  v2f z[16];
  auto load_fast = [&](int fb0) {
    const v2f* src = reinterpret_cast<const v2f*>(xr + ((int64_t)(fb0 + fl) * d.hop - shift)) + u;
#pragma unroll
    for (int n = 0; n < 16; ++n) z[n] = src[T * n];
  };
  bool fast = f_first < f_end && batch_fast(f_first);
  if (fast) load_fast(f_first);
  for (int g = 0; g < a.nbatch; ++g) {
    const int fb0 = f_first + g * FB;
    if (fb0 >= f_end) break;                                   // workgroup-uniform
    const int f = fb0 + fl;
    const int64_t p0 = (int64_t)f * d.hop;
    if (fast) {
      if (a.mul) {                                             // iSTFT backward: 1 / envelope per padded sample (wave-uniform branch)
        const float* mp = a.mul + p0 + 2 * u;
        v2f mv[16];
#pragma unroll
        for (int n = 0; n < 16; ++n) mv[n] = v2f{mp[2 * T * n], mp[2 * T * n + 1]};
#pragma unroll
        for (int n = 0; n < 16; ++n) z[n] = z[n] * mv[n];
      }
#pragma unroll
      for (int n = 0; n < 16; ++n) z[n] = z[n] * wv[n];
    } else {
#pragma unroll 1
      for (int n = 0; n < 16; ++n) {
        float s[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int t = 2 * (u + T * n) + c, pp = (int)p0 + t, wi = t - woff;
          const int sm = map_sample(d, pp);
          const bool v = (f < f_end) & (sm >= 0) & (wi >= 0) & (wi < d.win);
          const float xv = xr[v ? sm : 0] * a.window[v ? wi : 0] * d.scale;
          const float mv = a.mul ? a.mul[v ? pp : 0] : 1.f;
          s[c] = v ? xv * mv : 0.f;
        }
        data[fl * K::FS + u + T * n] = v2f{s[0], s[1]};
      }
      lds_barrier();
#pragma unroll
      for (int n = 0; n < 16; ++n) z[n] = data[fl * K::FS + u + T * n];
      lds_barrier();
    }
    fft_core<LOGN>(z, data, st);
    lds_barrier();
    fast = (g + 1 < a.nbatch) && (fb0 + FB < f_end) && batch_fast(fb0 + FB);
    if (fast) load_fast(fb0 + FB);                             // z is dead until the next batch: its loads fly over the epilogue
    // split step on bin pairs (k, NC - k):
    //   E = (A + conj B) / 2, P = (A - conj B) / 2 * e^{-i pi k / NC}:  X[k] = E - i P,  X[NC - k] = conj E - i conj P
    // frame-major spectrum: lanes along bins, every frame's bins are one contiguous run (full-line stores)
    if (d.mode == RFX_STFT_COMPLEX_FM) {
      for (int idx = tid; idx < K::NH * FB; idx += 256) {
        const int fl2 = idx / K::NH, k = idx - fl2 * K::NH;
        const int f2 = fb0 + fl2;
        if (f2 >= f_end) continue;
        const v2f* fr = data + fl2 * K::FS;
        const v2f A = fr[fft_phys3<LOGN>(k)];
        const v2f B = cconj(fr[fft_phys3<LOGN>((NC - k) & (NC - 1))]);
        const v2f E = (A + B) * 0.5f, D = (A - B) * 0.5f;
        const v2f P = cmul(D, tb.twH[k]);
        v2f X0 = v2f{E.x + P.y, E.y - P.x}, X1 = v2f{E.x - P.y, -E.y - P.x};
        if (d.herm) {
          if (k == 0) { X0.y = 0.f; X1.y = 0.f; }
          else { X0 = X0 * 2.f; X1 = X1 * 2.f; }
        }
        v2f* o = reinterpret_cast<v2f*>(a.out) + ((int64_t)row * FO + (f2 - d.frame0)) * d.bins;
        if (k < d.bins) o[k] = X0;
        if (k != NC - k && NC - k < d.bins) o[NC - k] = X1;
      }
      lds_barrier();
      continue;
    }
    // transposing epilogue, lanes along frames
    for (int idx = tid; idx < K::NH * FB; idx += 256) {
      const int k = idx / FB, fl2 = idx & (FB - 1);
      const int f2 = fb0 + fl2;
      if (f2 >= f_end) continue;
      const v2f* fr = data + fl2 * K::FS;
      const v2f A = fr[fft_phys3<LOGN>(k)];
      const v2f B = cconj(fr[fft_phys3<LOGN>((NC - k) & (NC - 1))]);
      const v2f E = (A + B) * 0.5f, D = (A - B) * 0.5f;
      const v2f P = cmul(D, tb.twH[k]);
      v2f X[2] = {v2f{E.x + P.y, E.y - P.x}, v2f{E.x - P.y, -E.y - P.x}};
      const int kk[2] = {k, NC - k};
      const int fo = f2 - d.frame0;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if (c == 1 && k == NC - k) break;
        const int kc = kk[c];
        if (kc >= d.bins) continue;
        v2f Xc = X[c];
        if (d.herm) {  // gradient of irfft: middle bins doubled, DC / Nyquist imaginary part dropped
          if (kc == 0 || kc == NC) Xc.y = 0.f;
          else Xc = Xc * 2.f;
        }
        const int64_t rb = (int64_t)row * d.bins + kc;
        switch (d.mode) {
          case RFX_STFT_COMPLEX:
            reinterpret_cast<v2f*>(a.out)[rb * FO + fo] = Xc;
            break;
          case RFX_STFT_CAC:
            a.out[((int64_t)row * 2 * d.bins + kc) * FO + fo] = Xc.x;
            a.out[((int64_t)row * 2 * d.bins + d.bins + kc) * FO + fo] = Xc.y;
            break;
          case RFX_STFT_MAG:
            a.out[rb * FO + fo] = sqrtf(fmaxf(Xc.x * Xc.x + Xc.y * Xc.y, d.eps));
            break;
          case RFX_STFT_POW:
            a.out[rb * FO + fo] = Xc.x * Xc.x + Xc.y * Xc.y;
            break;
          default:  // RFX_STFT_MAGPOW
            a.out[rb * FO + fo] = powf(sqrtf(Xc.x * Xc.x + Xc.y * Xc.y) + d.eps, d.alpha);
            break;
        }
      }
    }
    lds_barrier();      // the next batch's pass A / B writes E1 / E2 over this batch's E3
  }
}

// OWN = overlap-add by ownership (no atomics, no zero fill, bit-reproducible): the frames of a row are cut into WALKS of
// a.walk_frames frames; a workgroup runs the a.halo_frames = (win - 1) / hop frames in front of its walk first (they overlap its first
// samples; the previous walk transforms them too: ~10 % more transforms), keeps the not-yet-complete tail of the running sum in an LDS
// ring and stores a padded position as soon as no later frame can reach it -- positions in [F0 hop + woff, F1 hop + woff) of walk
// [F0, F1) are its own, so every output sample is written by exactly one thread of one workgroup.  The adjoint of the reflect-padded
// STFT (in_mode 0) folds the two edge zones of a row in LDS (each edge sample has exactly two contributions, direct and mirrored, so
// the order of the two LDS adds does not matter) and stores them at the end of the first / last walk.
template <int LOGN, bool LG = false, bool OWN = false>
__global__ __launch_bounds__(256, 3) void fft_synthesis_kernel(const FftArgs a) {
  typedef FftCfg<LOGN> K;
  constexpr int NC = K::NC, T = K::T, FB = K::FB, N = 2 * NC;
  __shared__ v2f data[FB * K::FS];
  __shared__ FftTables<LOGN> tb;
  const rfx_stft_desc& d = a.d;
  int row, f_first;
  block_coords<LOGN>(a, row, f_first);
  if (row >= d.R) return;
  const int tid = threadIdx.x, u = tid & (T - 1), fl = tid / T;
  FftState<LOGN> st;
  fft_setup<LOGN>(a, tb, st);
  lds_barrier();                       // the merge step below reads the table other threads loaded
  // The synthesis window is applied where the inverse transform leaves the registers (pass C): output m of a thread is time index
  // i = mA + RA (mB + 16 m) = samples (2 i, 2 i + 1); its 16 window pairs are frame-independent.  The overlap-add then only sums LDS
  // values: with the window fetched per (position, covering frame) its inner loop was a chain of global-load latencies.
  const int woff = (N - d.win) / 2;
  v2f wout[16];
  {
    float wl[16][2];
#pragma unroll
    for (int m = 0; m < 16; ++m)
#pragma unroll
      for (int c = 0; c < 2; ++c)
        wl[m][c] = a.window[min(max(2 * ((u >> 4) + K::RA * ((u & 15) + 16 * m)) + c - woff, 0), d.win - 1)];
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int w0 = 2 * ((u >> 4) + K::RA * ((u & 15) + 16 * m)) - woff;
      wout[m] = v2f{((w0 >= 0) & (w0 < d.win)) ? wl[m][0] : 0.f, ((w0 + 1 >= 0) & (w0 + 1 < d.win)) ? wl[m][1] : 0.f};
    }
  }
  const int f_end = d.frame0 + d.frames_out;
  const int FO = d.frames_out;
  float* outr = a.out + (int64_t)row * d.T;
  constexpr bool lg = LG;                                    // loss-gradient source (its own instantiation); the row's two weights
  float lg_ksc = 0.f, lg_wlm = 0.f;
  if (lg) {
    float wsc = a.lg_wsc;
    lg_wlm = a.lg_wlm;
    if (a.lg_gup) { const float uu = a.lg_gup[0]; wsc *= uu; lg_wlm *= uu; }
    const float A = a.lg_sums[3 * row], B = a.lg_sums[3 * row + 1];
    lg_ksc = (A > 0.f && B > 0.f) ? wsc / (sqrtf(A) * sqrtf(B)) : 0.f;
  }
  // OWN: walk w = frames [F0, F1) of the row, run-in from hs
  int F0 = 0, F1 = f_end, hs = f_first;
  int64_t own_lo = INT64_MIN, own_hi = INT64_MAX;
  // OWN scratch in global memory (LDS has no room: a third workgroup per CU is worth more than the carry's L2 round trip, measured):
  // the carry of this workgroup -- two buffers of N floats, written for the NEXT batch's positions while this batch's are read --
  // and the row's two edge zones as PADDED positions (folded by fft_fold_zones_kernel after the launch)
  float* carry = a.ws + (int64_t)d.R * 2 * (N + 1) + (int64_t)blockIdx.x * 2 * N;
  float* zoneL = a.ws + ((int64_t)row * 2) * (N + 1);
  float* zoneR = zoneL + (N + 1);
  int cin = 0;                                              // positions q < cin of the current batch have a carry from the previous one
  bool first_walk = true, last_walk = true;
  if (OWN) {
    const int w = (blockIdx.x >> 3) % a.groups_per_row;
    first_walk = w == 0; last_walk = w == a.groups_per_row - 1;
    F0 = d.frame0 + w * a.walk_frames;
    F1 = last_walk ? f_end : F0 + a.walk_frames;
    hs = max(d.frame0, F0 - a.halo_frames);
    if (!first_walk) own_lo = (int64_t)F0 * d.hop + woff;
    if (!last_walk) own_hi = (int64_t)F1 * d.hop + woff;
  }
  const int f_stop = OWN ? F1 : f_end;                      // frames at and beyond f_stop are not this workgroup's
  const bool fm = d.mode == RFX_STFT_COMPLEX_FM;           // frame-major spectrum: lanes along bins
  // all loads of the thread's NIT items first (clamped indices, validity applied to the values): a load -> use loop costs one
  // memory round trip per item and batch
  constexpr int NIT = (K::NH * FB + 255) / 256;
  v2f xkv[NIT], xmv[NIT];
  float ykv[NIT], ymv[NIT];                                // loss-gradient source: the target magnitudes of the same cells
  auto item = [&](int it, int& fl2, int& k) -> bool {
    const int idx = tid + 256 * it;
    const bool act = idx < K::NH * FB;
    const int ii = act ? idx : 0;
    fl2 = fm ? ii / K::NH : ii & (FB - 1);
    k = fm ? ii - fl2 * K::NH : ii / FB;
    return act;
  };
  auto fetch = [&](int kq, int fo) -> v2f {
    const int kc = kq < d.bins ? kq : 0;
    v2f v;
    if (fm) v = reinterpret_cast<const v2f*>(a.x)[((int64_t)row * FO + fo) * d.bins + kc];
    else if (d.mode == RFX_STFT_COMPLEX) v = reinterpret_cast<const v2f*>(a.x)[((int64_t)row * d.bins + kc) * FO + fo];   // uniform
    else {
      v.x = a.x[((int64_t)row * 2 * d.bins + kc) * FO + fo];
      v.y = a.x[((int64_t)row * 2 * d.bins + d.bins + kc) * FO + fo];
    }
    return v;
  };
  auto fix = [&](v2f v, int kq) -> v2f {
    if (kq >= d.bins) v = v2f{0.f, 0.f};
    if (kq == 0 || kq == NC) v.y = 0.f;             // real by construction / ignored by irfft
    else if (!d.herm) v = v * 0.5f;                 // adjoint of the one-sided rfft
    return v;
  };
  // all loads of a batch: the thread's NIT items (clamped indices, validity applied to the values)
  auto load_batch = [&](int fb) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      int fl2, k;
      item(it, fl2, k);
      const int f2 = fb + fl2;
      const int fo = f2 < f_stop ? f2 - d.frame0 : 0;
      xkv[it] = fetch(k, fo);
      xmv[it] = fetch(NC - k, fo);
      if (lg) {
        const int64_t cell0 = ((int64_t)row * FO + fo) * d.bins;
        ykv[it] = a.lg_ymag[cell0 + (k < d.bins ? k : 0)];
        ymv[it] = a.lg_ymag[cell0 + (NC - k < d.bins ? NC - k : 0)];
      }
    }
  };
  for (int g = 0; OWN || g < a.nbatch; ++g) {
    const int fb0 = OWN ? hs + g * FB : f_first + g * FB;
    if (fb0 >= f_stop) break;
    // merge step on bin pairs (k, NC - k), k in [0, NC / 2]: S = X[k] + conj X[NC-k], W = (X[k] - conj X[NC-k]) e^{+i pi k/NC}
    //   Z[k] = S + i W,  Z[NC - k] = conj S + i conj W;  conj Z is written in natural order (the inverse = conj FFT conj)
    load_batch(fb0);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      int fl2, k;
      if (!item(it, fl2, k)) continue;
      const bool fv = fb0 + fl2 < f_stop;
      if (LG) {                                               // one item at a time: unrolled together the reciprocals of all items spill
        xkv[it] = fft_lossgrad_cell(xkv[it], ykv[it], lg_ksc, lg_wlm, a.lg_eps);
        xmv[it] = fft_lossgrad_cell(xmv[it], ymv[it], lg_ksc, lg_wlm, a.lg_eps);
        __builtin_amdgcn_sched_barrier(0);
      }
      const v2f xk = fix(xkv[it], k), xm = cconj(fix(xmv[it], NC - k));
      v2f Z0 = v2f{0.f, 0.f}, Z1 = v2f{0.f, 0.f};
      if (fv) {
        const v2f S = xk + xm, D = xk - xm;
        const v2f W = cmul(D, cconj(tb.twH[k]));
        Z0 = v2f{S.x - W.y, -(S.y + W.x)};            // conj Z[k]
        Z1 = v2f{S.x + W.y, S.y - W.x};               // conj Z[NC - k]
      }
      v2f* fr = data + fl2 * K::FS;
      fr[k] = Z0;
      if (k != 0 && k != NC - k) fr[NC - k] = Z1;
    }
    lds_barrier();
    v2f z[16];
#pragma unroll
    for (int n = 0; n < 16; ++n) z[n] = data[fl * K::FS + u + T * n];
    lds_barrier();
    fft_core<LOGN>(z, data, st, wout);
    lds_barrier();
    const int nf = min(FB, f_stop - fb0);
    const int span = (nf - 1) * d.hop + N;
    const int64_t p0 = (int64_t)fb0 * d.hop;
    if (OWN) {
      // Every padded position q of the batch's span: the batch's frames that cover it (out of LDS, as below) + what earlier batches
      // carried over.  Complete (no later frame of the row starts at or before it) -> stored if it is this walk's own, else it is
      // carried to the next batch.  A position is one thread's per batch; the carry buffers alternate, so no position is read and
      // written in the same batch; the stores are drained (vmcnt) before the batch's barrier and read back past L1 (sc1).
      const int adv = nf * d.hop;
      const int fin = (fb0 + nf >= f_end) ? span : adv + woff;
      const float* cr = carry + (g & 1) * N;
      float* cw = carry + ((g + 1) & 1) * N;
      for (int q = tid; q < span; q += 256) {
        const int qw = q - woff;
        if (qw < 0) continue;                          // complete since the previous batch
        float v = q < cin ? __hip_atomic_load(cr + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
        const int lo_num = qw - d.win + 1;
        int fl_hi, fl_lo;
        if (a.hop_magic) {
          fl_hi = min(nf - 1, (int)__umulhi((uint32_t)qw, a.hop_magic));
          fl_lo = lo_num > 0 ? (int)__umulhi((uint32_t)(lo_num + d.hop - 1), a.hop_magic) : 0;
        } else {
          fl_hi = min(nf - 1, qw / d.hop);
          fl_lo = lo_num > 0 ? (lo_num + d.hop - 1) / d.hop : 0;
        }
        float gsum = 0.f;
        for (int f2 = fl_lo; f2 <= fl_hi; ++f2) {
          const int t = q - f2 * d.hop;
          const v2f zz = data[f2 * K::FS + fft_phys3<LOGN>(t >> 1)];
          gsum += (t & 1) ? -zz.y : zz.x;
        }
        v += gsum;
        if (q >= fin) { cw[q - adv] = v; continue; }
        const int64_t p = p0 + q;
        if (p < own_lo || p >= own_hi) continue;
        v *= d.scale;
        if (a.mul) v *= a.mul[p];
        if (d.in_mode == 1) {
          const int64_t sx = p - d.in_offset;
          if (sx >= 0 && sx < d.T) outr[sx] = d.accum ? outr[sx] + v : v;
        } else {                                        // adjoint of centre + reflect padding (no extra pads: the launcher checks)
          const int sx = (int)p - NC;                   // NC = n_fft / 2: the sample this padded position is, before mirroring
          if (sx <= NC) zoneL[p] = v;                   // padded positions [0, n_fft]: samples 0 .. NC, direct and mirrored
          else if (sx >= d.T - 1 - NC) { if (p - (d.T - 1) <= N) zoneR[p - (d.T - 1)] = v; }   // padded [T - 1, T - 1 + n_fft]
          else outr[sx] = d.accum ? outr[sx] + v : v;
        }
      }
      cin = span - adv;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
    // Overlap-add by GATHER inside the workgroup: the batch's frames are consecutive, so every padded position p of their span
    // sums the <= ceil(win / hop) frames that cover it out of LDS and issues ONE global atomic (the scatter form issued
    // one per (frame, sample): 3.3-4.3x more, and the loss-gradient launches ran at the L2 atomic rate, 0.46 TB/s).
    for (int q = tid; q < span; q += 256) {
      const int qw = q - woff;                       // window index of frame 0 at this position
      if (qw < 0) continue;
      const int lo_num = qw - d.win + 1;
      int fl_hi, fl_lo;
      if (a.hop_magic) {                             // wave-uniform
        fl_hi = min(nf - 1, (int)__umulhi((uint32_t)qw, a.hop_magic));
        fl_lo = lo_num > 0 ? (int)__umulhi((uint32_t)(lo_num + d.hop - 1), a.hop_magic) : 0;
      } else {
        fl_hi = min(nf - 1, qw / d.hop);
        fl_lo = lo_num > 0 ? (lo_num + d.hop - 1) / d.hop : 0;
      }
      if (fl_lo > fl_hi) continue;
      float v = 0.f;
      for (int f2 = fl_lo; f2 <= fl_hi; ++f2) {
        const int t = q - f2 * d.hop;
        const v2f zz = data[f2 * K::FS + fft_phys3<LOGN>(t >> 1)];
        v += (t & 1) ? -zz.y : zz.x;               // already windowed (pass C)
      }
      const int64_t p = p0 + q;
      const int sidx = map_sample(d, (int)p);
      if (sidx < 0) continue;
      v *= d.scale;
      if (a.mul) v *= a.mul[p];
      atomicAdd(outr + sidx, v);
    }
    }
    lds_barrier();
  }
}

// The edge zones of the reflect-padded adjoint (in_mode 0, OWN): sample s in [0, NC] = padded position NC + s plus its mirror image
// NC - s; sample T - 1 - NC + i = padded T - 1 + i plus the mirror image T - 1 + n_fft - i.  Positions no frame covers count 0.
// One thread per sample: two adds in a fixed order.
__global__ __launch_bounds__(256) void fft_fold_zones_kernel(const float* __restrict__ ws, float* __restrict__ out, int R, int T, int NC,
                                                             int64_t c_lo, int64_t c_hi, int accum) {
  const int per = 2 * (NC + 1);
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= R * per) return;
  const int row = idx / per, r = idx - row * per, side = r / (NC + 1), i = r - side * (NC + 1);
  const float* z = ws + ((int64_t)row * 2 + side) * (2 * NC + 1);
  const int64_t base = side ? (int64_t)T - 1 : 0;          // padded position of z[0]
  auto at = [&](int k) -> float { const int64_t p = base + k; return (p >= c_lo && p < c_hi) ? z[k] : 0.f; };
  float v;
  int s;
  if (side == 0) { s = i; v = at(NC + i); if (i >= 1) v += at(NC - i); }
  else { s = T - 1 - NC + i; v = at(i); if (i < NC) v += at(2 * NC - i); }
  float* o = out + (int64_t)row * T + s;
  *o = accum ? *o + v : v;
}

static bool stft_desc_ok(const rfx_stft_desc* d) {
  if (!d) return false;
  const int n = d->n_fft;
  if (n != 512 && n != 1024 && n != 2048 && n != 4096) return false;
  return d->R > 0 && d->T > 0 && d->hop > 0 && d->win > 0 && d->win <= n && d->frames_out > 0 &&
         d->bins > 0 && d->bins <= n / 2 + 1 && d->frame0 >= 0;
}

// twiddle tables of one n_fft on the current device: [256] W_256^t | [256] W_NC^t | [NC/2 + 1] e^{-i pi k / NC}
static const v2f* fft_tables(int nc) {
  static std::mutex mu;
  static v2f* tab[16][4] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  const int slot = nc == 256 ? 0 : nc == 512 ? 1 : nc == 1024 ? 2 : 3;
  std::lock_guard<std::mutex> lock(mu);
  if (tab[dev][slot]) return tab[dev][slot];
  const int n = 512 + nc / 2 + 1;
  float* h = new float[2 * n];
  const double pi = 3.14159265358979323846;
  for (int t = 0; t < 256; ++t) {
    h[2 * t] = (float)cos(-2.0 * pi * t / 256.0);
    h[2 * t + 1] = (float)sin(-2.0 * pi * t / 256.0);
    h[2 * (256 + t)] = (float)cos(-2.0 * pi * t / (double)nc);
    h[2 * (256 + t) + 1] = (float)sin(-2.0 * pi * t / (double)nc);
  }
  for (int k = 0; k <= nc / 2; ++k) {
    h[2 * (512 + k)] = (float)cos(-pi * k / (double)nc);
    h[2 * (512 + k) + 1] = (float)sin(-pi * k / (double)nc);
  }
  v2f* p = nullptr;
  if (hipMalloc(&p, sizeof(float) * 2 * n) == hipSuccess && hipMemcpy(p, h, sizeof(float) * 2 * n, hipMemcpyHostToDevice) == hipSuccess)
    tab[dev][slot] = p;
  else if (p) { (void)hipFree(p); p = nullptr; }
  delete[] h;
  return tab[dev][slot];
}

struct FftLossGradSrc {
  const float* ymag; const float* sums; const float* gup;
  float w_sc, w_lm, eps;
};
// Geometry of the ownership form of the synthesis (0 walks: the geometry takes the atomic form)
struct SynOwn { int walk_frames, halo, nwalks; unsigned grid; int64_t ws_floats; };
static SynOwn syn_own_geometry(const rfx_stft_desc* d) {
  SynOwn o{0, 0, 0, 0, 0};
  static const int own_on = [] { const char* e = getenv("RFX_FFT_OWN"); return e ? atoi(e) : 1; }();
  const int woff = (d->n_fft - d->win) / 2, nc = d->n_fft / 2, fb = 4096 / nc;
  const bool own = own_on && d->hop <= d->win &&
                   (d->in_mode == 1 || (!d->extra_pad_l && !d->extra_pad_r && d->frame0 == 0 && d->T > 2 * d->n_fft + 2));
  if (!own) return o;
  const int halo = (d->win - 1) / d->hop;
  int fw_min = halo + 1;
  if (d->in_mode == 0) {                                     // walk 0 owns padded [0, n_fft], the last walk padded [T - 1, ...)
    fw_min = std::max(fw_min, (d->n_fft + 1 - woff + d->hop - 1) / d->hop + 1);
    fw_min = std::max(fw_min, 2 + (woff + d->hop) / d->hop);
  }
  fw_min = std::max(fw_min, 2 * halo);                       // run-in <= half of a walk
  int64_t fw = ((int64_t)d->frames_out * d->R + 2047) / 2048; // ~2048 workgroups ...
  fw = std::min<int64_t>(fw, std::max(16 * halo, 4 * fb));    // ... of walks no longer than needed to make the run-in cheap
  static const int fw_env = [] { const char* e = getenv("RFX_FFT_OWN_FW"); return e ? atoi(e) : 0; }();      // dev: frames per walk
  if (fw_env > 0) fw = fw_env;
  fw = std::max<int64_t>(fw, fw_min);
  fw = (fw + fb - 1) / fb * fb;
  o.walk_frames = (int)fw; o.halo = halo;
  o.nwalks = std::max<int>(1, (int)(d->frames_out / fw));     // the last walk takes the remainder (< 2 fw frames)
  o.grid = (unsigned)((d->R + 7) / 8 * o.nwalks * 8);
  o.ws_floats = (int64_t)d->R * 2 * (d->n_fft + 1) + (int64_t)o.grid * 2 * d->n_fft;
  return o;
}
extern "C" int64_t rfx_fft_synthesis_ws(const rfx_stft_desc* d) {
  if (!stft_desc_ok(d)) return -1;
  const int64_t n = syn_own_geometry(d).ws_floats;
  return n > 0 ? n : 1;
}

template <bool SYN>
static int launch_fft(const rfx_stft_desc* d, const float* x, const float* window, const float* mul,
                      float* out, void* stream, const FftLossGradSrc* lgsrc = nullptr, float* ws = nullptr) {
  if (!stft_desc_ok(d) || !x || !window || !out) return -1;
  FftArgs a;
  a.d = *d; a.x = x; a.window = window; a.mul = mul; a.out = out;
  a.lg_ymag = a.lg_sums = a.lg_gup = nullptr;
  a.lg_wsc = a.lg_wlm = a.lg_eps = 0.f;
  if (lgsrc) {
    if (!SYN || d->mode != RFX_STFT_COMPLEX_FM || !lgsrc->ymag || !lgsrc->sums) return -1;
    a.lg_ymag = lgsrc->ymag; a.lg_sums = lgsrc->sums; a.lg_gup = lgsrc->gup;
    a.lg_wsc = lgsrc->w_sc; a.lg_wlm = lgsrc->w_lm; a.lg_eps = lgsrc->eps;
  }
  const int nc = d->n_fft / 2;
  a.tables = fft_tables(nc);
  if (!a.tables) return -3;
  a.hop_magic = d->hop <= 4096 ? (uint32_t)((0x100000000ULL + (uint64_t)d->hop - 1) / (uint64_t)d->hop) : 0u;   // exact for q * hop < 2^32
  const int fb = 4096 / nc;
  const int batches = (d->frames_out + fb - 1) / fb;
  const int rows8 = (d->R + 7) / 8;
  // frame batches per workgroup: amortises the per-workgroup setup (tables, window registers) while the grid stays several times
  // the 768 workgroups the chip holds at once.  Only for the frame-major layout: with a [bin][frame] spectrum the workgroups that
  // share a 128-byte line must run at the same time for their partial-line stores to merge in L2 (_spec: 253 us at one batch per
  // workgroup, 370 / 458 us at two / four -- measured).
  int nb = 1;
  static const int syn_nb_max = [] { const char* e = getenv("RFX_FFT_SYN_NB"); const int v = e ? atoi(e) : 2; return v < 1 ? 1 : (v > 8 ? 8 : v); }();
  static const int ana_nb_max = [] { const char* e = getenv("RFX_FFT_ANA_NB"); const int v = e ? atoi(e) : 4; return v < 1 ? 1 : (v > 16 ? 16 : v); }();
  if (d->mode == RFX_STFT_COMPLEX_FM)
    while (nb < (SYN ? syn_nb_max : ana_nb_max) && (int64_t)rows8 * 8 * ((batches + 2 * nb - 1) / (2 * nb)) >= 4096) nb *= 2;
  a.nbatch = nb;
  a.groups_per_row = (batches + nb - 1) / nb;
  a.walk_frames = a.halo_frames = 0; a.ws = nullptr; a.cover_lo = a.cover_hi = 0;
  unsigned grid = (unsigned)(rows8 * a.groups_per_row * 8);
  hipStream_t s = (hipStream_t)stream;
  if (SYN) {
    // Overlap-add by ownership wherever the geometry allows it (everything the networks and losses use); else the atomic form into
    // a zero-filled output (extra reflect pads = the backward of HDemucs' _spec, which nothing differentiates; rows too short for two
    // disjoint edge zones; hop > win).
    const int woff = (d->n_fft - d->win) / 2;
    const SynOwn own = syn_own_geometry(d);
    if (own.nwalks) {
      if (!ws) return -1;
      a.walk_frames = own.walk_frames; a.halo_frames = own.halo; a.groups_per_row = own.nwalks; a.ws = ws;
      grid = own.grid;
      // every sample stored?  covered padded positions: [frame0 hop + woff, (frame0 + frames_out - 1) hop + woff + win)
      const int64_t c_lo = (int64_t)d->frame0 * d->hop + woff, c_hi = (int64_t)(d->frame0 + d->frames_out - 1) * d->hop + woff + d->win;
      const bool full = d->in_mode == 1 ? (c_lo <= d->in_offset && c_hi >= (int64_t)d->in_offset + d->T)
                                        : (c_lo <= 2 * (int64_t)nc + 1 && c_hi >= (int64_t)d->T - 1);
      if (!full && !d->accum && hipMemsetAsync(out, 0, sizeof(float) * (size_t)d->R * d->T, s) != hipSuccess) return -3;
      a.cover_lo = c_lo; a.cover_hi = c_hi;
#define RFX_SYN_OWN(LOGN, LGV) hipLaunchKernelGGL((fft_synthesis_kernel<LOGN, LGV, true>), dim3(grid), dim3(256), 0, s, a)
      if (lgsrc) {
        switch (d->n_fft) {
          case 512: RFX_SYN_OWN(8, true); break;
          case 1024: RFX_SYN_OWN(9, true); break;
          case 2048: RFX_SYN_OWN(10, true); break;
          default: RFX_SYN_OWN(11, true); break;
        }
      } else {
        switch (d->n_fft) {
          case 512: RFX_SYN_OWN(8, false); break;
          case 1024: RFX_SYN_OWN(9, false); break;
          case 2048: RFX_SYN_OWN(10, false); break;
          default: RFX_SYN_OWN(11, false); break;
        }
      }
#undef RFX_SYN_OWN
      RFX_CHECK_LAUNCH();
      if (d->in_mode == 0) {
        const int n = d->R * 2 * (nc + 1);
        hipLaunchKernelGGL(fft_fold_zones_kernel, dim3((n + 255) / 256), dim3(256), 0, s, ws, out, d->R, d->T, nc, c_lo, c_hi, d->accum);
        RFX_CHECK_LAUNCH();
      }
      return 0;
    }
    if (!d->accum && hipMemsetAsync(out, 0, sizeof(float) * (size_t)d->R * d->T, s) != hipSuccess) return -3;   // atomic form
  }
  if (SYN && lgsrc) {
    switch (d->n_fft) {
      case 512: hipLaunchKernelGGL((fft_synthesis_kernel<8, true>), dim3(grid), dim3(256), 0, s, a); break;
      case 1024: hipLaunchKernelGGL((fft_synthesis_kernel<9, true>), dim3(grid), dim3(256), 0, s, a); break;
      case 2048: hipLaunchKernelGGL((fft_synthesis_kernel<10, true>), dim3(grid), dim3(256), 0, s, a); break;
      default: hipLaunchKernelGGL((fft_synthesis_kernel<11, true>), dim3(grid), dim3(256), 0, s, a); break;
    }
    RFX_CHECK_LAUNCH();
    return 0;
  }
  switch (d->n_fft) {
    case 512:
      if (SYN) hipLaunchKernelGGL(fft_synthesis_kernel<8>, dim3(grid), dim3(256), 0, s, a);
      else hipLaunchKernelGGL(fft_analysis_kernel<8>, dim3(grid), dim3(256), 0, s, a);
      break;
    case 1024:
      if (SYN) hipLaunchKernelGGL(fft_synthesis_kernel<9>, dim3(grid), dim3(256), 0, s, a);
      else hipLaunchKernelGGL(fft_analysis_kernel<9>, dim3(grid), dim3(256), 0, s, a);
      break;
    case 2048:
      if (SYN) hipLaunchKernelGGL(fft_synthesis_kernel<10>, dim3(grid), dim3(256), 0, s, a);
      else hipLaunchKernelGGL(fft_analysis_kernel<10>, dim3(grid), dim3(256), 0, s, a);
      break;
    default:
      if (SYN) hipLaunchKernelGGL(fft_synthesis_kernel<11>, dim3(grid), dim3(256), 0, s, a);
      else hipLaunchKernelGGL(fft_analysis_kernel<11>, dim3(grid), dim3(256), 0, s, a);
      break;
  }
  RFX_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------
// MR-STFT loss forward in one launch per resolution (auraloss STFTLoss terms behind models.py:320).  TWO FRAMES of one signal
// come out of ONE n_fft-point complex FFT of z = w s_a + i w s_b (s_a, s_b = the samples of frames f and f + FB):
//   S_a[k] = (Z[k] + conj Z[N-k]) / 2,   S_b[k] = -i (Z[k] - conj Z[N-k]) / 2,   k = 0 .. N/2.
// A batch transforms 2 FB frames of the prediction x, keeps their clamped powers in registers, transforms the same frames of
// the target y with the SAME instruction sequence (identical signals give bit-identical spectra: loss(x, x) = 0 exactly, as
// with two torch.stft calls) and takes the three row sums { sum (|Y| - |X|)^2, sum |Y|^2, sum |log|X| - log|Y|| } in the
// epilogue: neither spectrum is written unless the backward needs them (xspec: X frame-major complex, ymag: clamped |Y|), and
// the separate reduction pass over both spectra (1.3 GB read per launch at B = 64) is gone.  Same passes / layouts as the kernels
// above with NC = n_fft; a thread's 16 points are sample pairs of its two frames, its window values scalars.
// ---------------------------------------------------------------------------------
struct PairArgs {
  rfx_stft_desc d;        // R, T, n_fft, hop, win, frames_out (all frames), bins = n_fft/2 + 1, in_mode 0
  const float* x;
  const float* y;
  const float* window;
  const v2f* tables;
  double* slots;          // [R][groups_per_row][3]: one slot per workgroup
  v2f* xspec;             // optional [R][frames][bins]
  float* ymag;            // optional [R][frames][bins]
  float eps;
  int groups_per_row, nbatch;
};

template <int LOGN>
__global__ __launch_bounds__(256, (LOGN == 9 ? 2 : RFX_FFT_PAIR_OCC)) void fft_pair_loss_kernel(const PairArgs a) {
  typedef FftCfg<LOGN> K;
  constexpr int NC = K::NC, T = K::T, FB = K::FB, N = NC;
  constexpr int NIT = (K::NH * FB + 255) / 256;                 // epilogue items per thread
  __shared__ v2f data[FB * K::FS];
  __shared__ FftTables<LOGN> tb;
  __shared__ double part[3][4];
  const rfx_stft_desc& d = a.d;
  const int b = blockIdx.x, xcd = b & 7, qb = b >> 3;
  const int slot = qb / a.groups_per_row, grp = qb - slot * a.groups_per_row;
  const int row = slot * 8 + xcd;
  if (row >= d.R) return;
  const int f_first = grp * (2 * FB * a.nbatch);
  const int tid = threadIdx.x, u = tid & (T - 1), fl = tid / T;
  FftState<LOGN> st;
  FftArgs fa;
  fa.tables = a.tables;
  fft_setup<LOGN>(fa, tb, st);
  lds_barrier();
  const int f_end = d.frames_out;
  const float* xr = a.x + (int64_t)row * d.T;
  const float* yr = a.y + (int64_t)row * d.T;
  const int woff = (N - d.win) / 2;
  float wv[16];
  {
    float wl[16];
#pragma unroll
    for (int n = 0; n < 16; ++n) wl[n] = a.window[min(max(u + T * n - woff, 0), d.win - 1)];
#pragma unroll
    for (int n = 0; n < 16; ++n) {
      const int w0 = u + T * n - woff;
      wv[n] = ((w0 >= 0) & (w0 < d.win)) ? wl[n] : 0.f;
    }
  }
  const int shift = d.n_fft / 2;
  // workgroup-uniform: all 2 FB frames of the batch inside the signal -> plain loads, issued before the preceding epilogue
  auto batch_fast = [&](int fb0) -> bool {
    const int64_t pb0 = (int64_t)fb0 * d.hop;
    return (fb0 + 2 * FB <= f_end) & (pb0 - shift >= 0) & (pb0 + (int64_t)(2 * FB - 1) * d.hop + N - 1 - shift < (int64_t)d.T);
  };
  v2f z[16];
  auto load_fast = [&](const float* sr, int fb0) {
    const int64_t o = (int64_t)(fb0 + fl) * d.hop - shift + u;
    const int64_t ob = o + (int64_t)FB * d.hop;
#pragma unroll
    for (int n = 0; n < 16; ++n) z[n] = v2f{sr[o + T * n], sr[ob + T * n]};
  };
  // one signal's 2 FB frames: window (fast) or map + stage (edges), three passes -> Z in layout E3
  auto transform = [&](const float* sr, int fb0, bool fast) {
    if (fast) {
#pragma unroll
      for (int n = 0; n < 16; ++n) z[n] = z[n] * wv[n];
    } else {
#pragma unroll 1
      for (int n = 0; n < 16; ++n) {
        const int t = u + T * n, wi = t - woff;
        const bool wok = (wi >= 0) & (wi < d.win);
        const float wq = a.window[wok ? wi : 0];
        float sv[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int f = fb0 + fl + c * FB;
          const int sm = map_sample(d, f * d.hop + t);
          const bool v = (f < f_end) & (sm >= 0) & wok;
          const float q = sr[v ? sm : 0];
          sv[c] = v ? q * wq : 0.f;
        }
        data[fl * K::FS + t] = v2f{sv[0], sv[1]};
      }
      lds_barrier();
#pragma unroll
      for (int n = 0; n < 16; ++n) z[n] = data[fl * K::FS + u + T * n];
      lds_barrier();
    }
    fft_core<LOGN>(z, data, st);
    lds_barrier();
  };
  bool fast = f_first < f_end && batch_fast(f_first);
  if (fast) load_fast(xr, f_first);
  double accA = 0.0, accB = 0.0, accC = 0.0;
  for (int g = 0; g < a.nbatch; ++g) {
    const int fb0 = f_first + g * 2 * FB;
    if (fb0 >= f_end) break;                                   // workgroup-uniform
    transform(xr, fb0, fast);
    if (fast) load_fast(yr, fb0);                              // the target's loads fly over the prediction's epilogue
    float px[NIT][2];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      int idx = tid + 256 * it;
      asm volatile("" : "+v"(idx));                            // no index / address values carried from one epilogue to the other
      const bool act = idx < K::NH * FB;
      const int fl2 = act ? idx / K::NH : 0, k = act ? idx - fl2 * K::NH : 0;
      const v2f* fr = data + fl2 * K::FS;
      const v2f A = fr[fft_phys3<LOGN>(k)];
      const v2f Bc = cconj(fr[fft_phys3<LOGN>((N - k) & (N - 1))]);
      const v2f Sa = (A + Bc) * 0.5f, Dm = (A - Bc) * 0.5f;
      const v2f Sb = v2f{Dm.y, -Dm.x};                            // S_b = -i Dm
      px[it][0] = fmaxf(rfx_pow2(Sa.x, Sa.y), a.eps);
      px[it][1] = fmaxf(rfx_pow2(Sb.x, Sb.y), a.eps);
      if (a.xspec && act) {
        const int f2 = fb0 + fl2;
        if (f2 < f_end) a.xspec[((int64_t)row * d.frames_out + f2) * d.bins + k] = Sa;
        if (f2 + FB < f_end) a.xspec[((int64_t)row * d.frames_out + f2 + FB) * d.bins + k] = Sb;
      }
      __builtin_amdgcn_sched_barrier(0);                       // one item at a time: interleaving all NIT costs ~120 spilled VGPRs
    }
    lds_barrier();
    transform(yr, fb0, fast);
    fast = (g + 1 < a.nbatch) && (fb0 + 2 * FB < f_end) && batch_fast(fb0 + 2 * FB);
    if (fast) load_fast(xr, fb0 + 2 * FB);
    float fa0 = 0.f, fb1 = 0.f, fc2 = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      int idx = tid + 256 * it;
      asm volatile("" : "+v"(idx));                            // no index / address values carried from one epilogue to the other
      const bool act = idx < K::NH * FB;
      const int fl2 = act ? idx / K::NH : 0, k = act ? idx - fl2 * K::NH : 0;
      const v2f* fr = data + fl2 * K::FS;
      const v2f A = fr[fft_phys3<LOGN>(k)];
      const v2f Bc = cconj(fr[fft_phys3<LOGN>((N - k) & (N - 1))]);
      const v2f Sa = (A + Bc) * 0.5f, Dm = (A - Bc) * 0.5f;
      const float py[2] = {fmaxf(rfx_pow2(Sa.x, Sa.y), a.eps), fmaxf(rfx_pow2(Dm.y, -Dm.x), a.eps)};
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int f2 = fb0 + fl2 + c * FB;
        const bool ok = act & (f2 < f_end);
        const float ym = __builtin_amdgcn_sqrtf(py[c]);
        const float dd = ym - __builtin_amdgcn_sqrtf(px[it][c]);
        fa0 += ok ? dd * dd : 0.f;
        fb1 += ok ? py[c] : 0.f;
        fc2 += ok ? fabsf(__builtin_amdgcn_logf(px[it][c]) - __builtin_amdgcn_logf(py[c])) : 0.f;
        if (a.ymag && ok) a.ymag[((int64_t)row * d.frames_out + f2) * d.bins + k] = ym;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    accA += (double)fa0; accB += (double)fb1; accC += (double)(fc2 * 0.34657359027997264f);   // log2 -> ln, halved
    lds_barrier();
  }
  accA = rfx_wave_sum_d(accA); accB = rfx_wave_sum_d(accB); accC = rfx_wave_sum_d(accC);
  const int lane = tid & 63, wave = tid >> 6;
  if (lane == 0) { part[0][wave] = accA; part[1][wave] = accB; part[2][wave] = accC; }
  lds_barrier();
  // the (row, frame group) slot is this workgroup's alone: stored, then added in group order by rfx_slot_sum_kernel (no fill, no atomics)
  if (tid < 3) a.slots[((int64_t)row * a.groups_per_row + grp) * 3 + tid] = (part[tid][0] + part[tid][1]) + (part[tid][2] + part[tid][3]);
}

static void pair_loss_geometry(const rfx_stft_desc* d, int& nb, int& groups) {
  const int fb = 2 * (4096 / d->n_fft);                        // frames per batch (n_fft complex points per pair transform)
  const int batches = (d->frames_out + fb - 1) / fb;
  const int rows8 = (d->R + 7) / 8;
  nb = 1;
  while (nb < 4 && (int64_t)rows8 * 8 * ((batches + 2 * nb - 1) / (2 * nb)) >= 4096) nb *= 2;
  groups = (batches + nb - 1) / nb;
}
// doubles of workspace rfx_stft_pair_loss needs for this geometry (3 per row and workgroup)
extern "C" int64_t rfx_stft_pair_loss_ws(const rfx_stft_desc* d) {
  if (!stft_desc_ok(d) || d->n_fft > 2048) return -1;
  int nb, groups;
  pair_loss_geometry(d, nb, groups);
  return (int64_t)3 * d->R * groups;
}
extern "C" int rfx_stft_pair_loss(const rfx_stft_desc* d, const float* x, const float* y, const float* window, float eps,
                                  double* ws, float* sums, float* xspec, float* ymag, void* stream) {
  if (!stft_desc_ok(d) || !x || !y || !window || !sums || !ws || (xspec == nullptr) != (ymag == nullptr)) return -1;
  if (d->n_fft > 2048 || d->in_mode != 0 || d->extra_pad_l || d->extra_pad_r || d->frame0 != 0 || d->bins != d->n_fft / 2 + 1) return -1;
  PairArgs a;
  a.d = *d; a.x = x; a.y = y; a.window = window; a.slots = ws; a.eps = eps;
  a.xspec = reinterpret_cast<v2f*>(xspec); a.ymag = ymag;
  const int nc = d->n_fft;                                     // complex points of the pair transform
  a.tables = fft_tables(nc);
  if (!a.tables) return -3;
  const int rows8 = (d->R + 7) / 8;
  pair_loss_geometry(d, a.nbatch, a.groups_per_row);
  const unsigned grid = (unsigned)(rows8 * a.groups_per_row * 8);
  hipStream_t s = (hipStream_t)stream;
  switch (d->n_fft) {
    case 512: hipLaunchKernelGGL(fft_pair_loss_kernel<9>, dim3(grid), dim3(256), 0, s, a); break;
    case 1024: hipLaunchKernelGGL(fft_pair_loss_kernel<10>, dim3(grid), dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL(fft_pair_loss_kernel<11>, dim3(grid), dim3(256), 0, s, a); break;
  }
  RFX_CHECK_LAUNCH();
  hipLaunchKernelGGL(rfx_slot_sum_kernel<float>, RFX_SLOT_SUM_GRID(3 * d->R), 0, s, ws, d->R, a.groups_per_row, 3, sums);
  RFX_CHECK_LAUNCH();
  return 0;
}

extern "C" int rfx_fft_analysis(const rfx_stft_desc* d, const float* x, const float* window,
                                const float* mul, float* out, void* stream) {
  return launch_fft<false>(d, x, window, mul, out, stream);
}
extern "C" int rfx_fft_synthesis(const rfx_stft_desc* d, const float* spec, const float* window,
                                 const float* mul, float* ws, float* out, void* stream) {
  if (d && d->mode != RFX_STFT_COMPLEX && d->mode != RFX_STFT_CAC && d->mode != RFX_STFT_COMPLEX_FM) return -1;
  return launch_fft<true>(d, spec, window, mul, out, stream, nullptr, ws);
}
// rfx_stft_loss_grad_m + rfx_fft_synthesis in one launch (the backward of one auraloss STFTLoss resolution behind models.py:320):
// the gradient spectrum is never written -- the merge step computes it from the stored prediction spectrum and target magnitudes
extern "C" int rfx_fft_synthesis_lossgrad(const rfx_stft_desc* d, const float* xspec, const float* ymag, const float* sums, float w_sc,
                                          float w_lm, float eps, const float* gup, const float* window, float* ws, float* out,
                                          void* stream) {
  if (!d || d->mode != RFX_STFT_COMPLEX_FM || !ymag || !sums) return -1;
  FftLossGradSrc src{ymag, sums, gup, w_sc, w_lm, eps};
  return launch_fft<true>(d, xspec, window, nullptr, out, stream, &src, ws);
}
