// Framed real FFT kernels: STFT / iSTFT forward and backward for every geometry
// of the RemFX hot path (n_fft 512..4096, arbitrary hop, hann window of `win`
// samples centred in n_fft, centre + reflect padding).
//
// Replaces torch.stft / torch.istft call sites: utils.py:148-154 (spectrogram),
// HDemucs _spec/_ispec (models.py:319), auraloss STFTLoss (models.py:320 ...),
// Open-Unmix Separator (models.py:298), MelSpectrogram (classifier.py:200).
//
// Structure (one workgroup = 256 threads = FB consecutive frames of one row):
//   * a real n_fft-point transform is done as an NC = n_fft/2 point complex FFT of
//     z[n] = x[2n] + i x[2n+1] plus the standard split/merge step;
//   * the NC-point FFT runs IN PLACE in LDS: radix-4 decimation-in-frequency passes
//     (+ one radix-2 pass when log2 NC is odd), natural-order input -> digit-reversed
//     output; the inverse runs the mirrored decimation-in-time passes;
//   * FB * NC = RFX_FFT_PTS complex points (2048 / 4096, see below) + an NC-entry twiddle table per
//     workgroup; frames are padded by one element so that the transposing epilogue
//     (lanes along frames -> coalesced [bin][frame] stores) is bank-conflict free;
//   * workgroups that write the same (row, bin) lines are placed on the same XCD
//     (block b runs on XCD b % 8) so partial-line stores merge in one L2.
#include "common.h"

// complex points per workgroup: 2048 for n_fft <= 1024, 4096 above (16-32 KB of LDS).  Round 1 used 8192 (64 KB: two workgroups =
// 8 waves per CU between workgroup barriers) and the kernels were OCCUPANCY-bound: analysis 0.55 -> 0.32 ms, synthesis 1.1 -> 0.5 ms
// per launch with the smaller groups, at the price of shorter store runs in the transposing epilogue.
#ifndef RFX_FFT_PTS_SMALL
#define RFX_FFT_PTS_SMALL 2048
#endif
#ifndef RFX_FFT_PTS_LARGE
#define RFX_FFT_PTS_LARGE 4096
#endif
#define RFX_FFT_PTS(LOGN) ((LOGN) <= 9 ? RFX_FFT_PTS_SMALL : RFX_FFT_PTS_LARGE)
static int rfx_fft_pts(int nc) { return nc <= 512 ? RFX_FFT_PTS_SMALL : RFX_FFT_PTS_LARGE; }

struct FftArgs {
  rfx_stft_desc d;
  const float* x;       // analysis: signal [R][T];   synthesis: spectrum
  const float* window;  // [win]
  const float* mul;     // optional per padded-sample multiplier (iSTFT 1/envelope)
  float* out;           // analysis: spectrum;       synthesis: signal [R][T] (atomic accumulate)
  int groups_per_row;
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) {  // a * conj(b)
  return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}

template <int LOGN>
__device__ __forceinline__ int digit_pos(int k) {
  int pos = 0, L = 1 << LOGN, kk = k;
#pragma unroll
  for (int s = 0; s < LOGN / 2; ++s) {
    pos += (kk & 3) * (L >> 2);
    kk >>= 2;
    L >>= 2;
  }
  if (LOGN & 1) pos += (kk & 1);
  return pos;
}

// In-place FFT of FB frames of NC complex points each (frame stride FS).
template <int LOGN, bool INV>
__device__ __forceinline__ void fft_passes(float2* data, const float2* tw, int nframes_elems /*FB*NC*/) {
  constexpr int NC = 1 << LOGN, FS = NC + 1;
  const int tid = threadIdx.x;
  if (!INV) {
    for (int L = NC; L >= 4; L >>= 2) {
      const int q = L >> 2, tstep = NC / L;
      for (int idx = tid; idx < nframes_elems / 4; idx += 256) {
        const int fr = idx / (NC / 4), r = idx - fr * (NC / 4);
        const int gI = r / q, j = r - gI * q;
        float2* p = data + fr * FS + gI * L + j;
        const float2 x0 = p[0], x1 = p[q], x2 = p[2 * q], x3 = p[3 * q];
        const float2 t0 = make_float2(x0.x + x2.x, x0.y + x2.y), t1 = make_float2(x0.x - x2.x, x0.y - x2.y);
        const float2 t2 = make_float2(x1.x + x3.x, x1.y + x3.y);
        const float2 t3 = make_float2(x1.y - x3.y, -(x1.x - x3.x));  // (x1-x3) * (-i)
        p[0] = make_float2(t0.x + t2.x, t0.y + t2.y);
        p[q] = cmul(make_float2(t1.x + t3.x, t1.y + t3.y), tw[j * tstep]);
        p[2 * q] = cmul(make_float2(t0.x - t2.x, t0.y - t2.y), tw[2 * j * tstep]);
        p[3 * q] = cmul(make_float2(t1.x - t3.x, t1.y - t3.y), tw[3 * j * tstep]);
      }
      __syncthreads();
    }
    if (LOGN & 1) {
      for (int idx = tid; idx < nframes_elems / 2; idx += 256) {
        const int fr = idx / (NC / 2), r = idx - fr * (NC / 2);
        float2* p = data + fr * FS + 2 * r;
        const float2 a = p[0], b = p[1];
        p[0] = make_float2(a.x + b.x, a.y + b.y);
        p[1] = make_float2(a.x - b.x, a.y - b.y);
      }
      __syncthreads();
    }
  } else {
    if (LOGN & 1) {
      for (int idx = tid; idx < nframes_elems / 2; idx += 256) {
        const int fr = idx / (NC / 2), r = idx - fr * (NC / 2);
        float2* p = data + fr * FS + 2 * r;
        const float2 a = p[0], b = p[1];
        p[0] = make_float2(a.x + b.x, a.y + b.y);
        p[1] = make_float2(a.x - b.x, a.y - b.y);
      }
      __syncthreads();
    }
    for (int L = (LOGN & 1) ? 8 : 4; L <= NC; L <<= 2) {
      const int q = L >> 2, tstep = NC / L;
      for (int idx = tid; idx < nframes_elems / 4; idx += 256) {
        const int fr = idx / (NC / 4), r = idx - fr * (NC / 4);
        const int gI = r / q, j = r - gI * q;
        float2* p = data + fr * FS + gI * L + j;
        const float2 u0 = p[0];
        const float2 u1 = cmulc(p[q], tw[j * tstep]);
        const float2 u2 = cmulc(p[2 * q], tw[2 * j * tstep]);
        const float2 u3 = cmulc(p[3 * q], tw[3 * j * tstep]);
        const float2 t0 = make_float2(u0.x + u2.x, u0.y + u2.y), t1 = make_float2(u0.x - u2.x, u0.y - u2.y);
        const float2 t2 = make_float2(u1.x + u3.x, u1.y + u3.y);
        const float2 t3 = make_float2(-(u1.y - u3.y), u1.x - u3.x);  // (u1-u3) * (+i)
        p[0] = make_float2(t0.x + t2.x, t0.y + t2.y);
        p[q] = make_float2(t1.x + t3.x, t1.y + t3.y);
        p[2 * q] = make_float2(t0.x - t2.x, t0.y - t2.y);
        p[3 * q] = make_float2(t1.x - t3.x, t1.y - t3.y);
      }
      __syncthreads();
    }
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
  constexpr int FB = RFX_FFT_PTS(LOGN) >> LOGN;
  // same-row frame groups on the same XCD (block b -> XCD b % 8), adjacent in time
  const int b = blockIdx.x, xcd = b & 7, q = b >> 3;
  const int slot = q / a.groups_per_row, grp = q - slot * a.groups_per_row;
  row = slot * 8 + xcd;
  f_first = a.d.frame0 + grp * FB;
}

// e^{-i pi k / NC} for the split (analysis) / merge (synthesis) step: a table instead of one sincospif per output
// element (n_fft <= 2048; the 4096-point kernels keep sincospif, their LDS is full)
template <int LOGN>
__device__ __forceinline__ void build_half_twiddles(float2* tw2) {
  constexpr int NC = 1 << LOGN;
  if (LOGN > 10) return;
  for (int t = threadIdx.x; t < NC / 2; t += 256) {
    float s, c;
    sincospif(-(float)(2 * t + 1) / (float)NC, &s, &c);
    tw2[t] = make_float2(c, s);
  }
}
// k in [0, NC]: even k -> tw[k/2] = e^{-2 pi i (k/2) / NC}, odd k -> tw2[(k-1)/2]
template <int LOGN>
__device__ __forceinline__ float2 half_twiddle(const float2* tw, const float2* tw2, int k) {
  constexpr int NC = 1 << LOGN;
  if (LOGN <= 10) return ((k & 1) ? tw2 : tw)[k >> 1];
  float s, c;
  sincospif((float)k / (float)NC, &s, &c);
  return make_float2(c, -s);
}

// digit-reversed position of every natural index: one LDS read instead of ~12 integer instructions per use in the split / merge
// steps (the r02c profile has these kernels VALU-issue-bound: ~6000 instructions per thread, two thirds of them index arithmetic)
template <int LOGN>
__device__ __forceinline__ void build_rev(uint16_t* rev) {
  constexpr int NC = 1 << LOGN;
  for (int t = threadIdx.x; t < NC; t += 256) rev[t] = (uint16_t)digit_pos<LOGN>(t);
}

template <int LOGN>
__device__ __forceinline__ void build_twiddles(float2* tw) {
  constexpr int NC = 1 << LOGN;
  for (int t = threadIdx.x; t < NC; t += 256) {
    float s, c;
    sincospif(-2.0f * (float)t / (float)NC, &s, &c);
    tw[t] = make_float2(c, s);
  }
}

template <int LOGN>
__global__ __launch_bounds__(256) void fft_analysis_kernel(const FftArgs a) {
  constexpr int NC = 1 << LOGN, FB = RFX_FFT_PTS(LOGN) >> LOGN, FS = NC + 1, N = 2 * NC;
  __shared__ float2 data[FB * FS];
  __shared__ float2 tw[NC];
  __shared__ float2 tw2[LOGN <= 10 ? NC / 2 : 1];     // e^{-i pi k / NC} for ODD k (even k is tw[k/2]); LDS budget: n_fft <= 2048
  __shared__ uint16_t rev[NC];
  const rfx_stft_desc& d = a.d;
  int row, f_first;
  block_coords<LOGN>(a, row, f_first);
  if (row >= d.R) return;
  const int tid = threadIdx.x;
  build_twiddles<LOGN>(tw);
  build_half_twiddles<LOGN>(tw2);
  build_rev<LOGN>(rev);
  const int f_end = d.frame0 + d.frames_out;
  const float* xr = a.x + (int64_t)row * d.T;
  const int woff = (N - d.win) / 2;
  // Load phase, 4 complex points (8 samples) per thread and round: every index is clamped and every load is
  // unconditional, validity is applied to the VALUE afterwards.  (With the loads inside `if (in window) if (in signal)`
  // hipcc emitted one load + s_waitcnt vmcnt(0) per sample: 64 dependent memory round trips per thread and workgroup,
  // which is where the r01 kernels spent their time: 0.87 ms for 3.4 GFLOP.)
  // r02: the kernels are VALU-issue-bound, so (a) a thread's points have only NC / 256 distinct in-frame indices: their window
  // values (x scale, 0 outside the window) are fetched once into registers; (b) a workgroup whose whole span lies inside the
  // signal (all but the first / last few of a row) skips the reflect arithmetic of map_sample.
  constexpr int NCB = NC / 256;                                   // distinct i per thread (1, 2, 4, 8)
  float wv[NCB <= 4 ? NCB : 1][2];
  if (NCB <= 4) {
#pragma unroll
    for (int c = 0; c < (NCB <= 4 ? NCB : 1); ++c)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int wi = 2 * (tid + 256 * c) + u - woff;
        const bool wok = (wi >= 0) & (wi < d.win);
        wv[c][u] = wok ? a.window[wok ? wi : 0] * d.scale : 0.f;
      }
  }
  const int shift = d.in_mode == 0 ? d.n_fft / 2 + d.extra_pad_l : d.in_offset;
  const int64_t p_lo = (int64_t)f_first * d.hop, p_hi = (int64_t)(f_first + FB - 1) * d.hop + N - 1;
  const bool interior = (f_first + FB <= f_end) & (p_lo - shift >= 0) & (p_hi - shift < (int64_t)d.T);      // workgroup-uniform
  for (int j0 = 0; j0 < (FB * NC) / 256; j0 += 4) {
    float xs[8], ws[8], ms[8];
    bool ok[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + 256 * (j0 + e);
      const int fl = idx >> LOGN, i = idx & (NC - 1);
      const int f = f_first + fl;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = 2 * i + u;
        const int pp = f * d.hop + t;
        if (NCB <= 4) {
          const int sm = interior ? pp - shift : map_sample(d, pp);
          const bool v = interior | ((f < f_end) & (sm >= 0));
          ok[2 * e + u] = v;
          xs[2 * e + u] = xr[v ? sm : 0];
          ws[2 * e + u] = wv[e & (NCB <= 4 ? NCB - 1 : 0)][u];      // j0 is a multiple of 4: (j0 + e) mod NCB = e mod NCB
          ms[2 * e + u] = a.mul ? a.mul[v ? pp : 0] : 1.f;       // a.mul: wave-uniform
        } else {
          const int wi = t - woff;
          const int sm = map_sample(d, pp);
          const bool v = (f < f_end) & (wi >= 0) & (wi < d.win) & (sm >= 0);
          ok[2 * e + u] = v;
          xs[2 * e + u] = xr[v ? sm : 0];
          ws[2 * e + u] = a.window[v ? wi : 0] * d.scale;
          ms[2 * e + u] = a.mul ? a.mul[v ? pp : 0] : 1.f;
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + 256 * (j0 + e);
      const int fl = idx >> LOGN, i = idx & (NC - 1);
      float2 v;
      v.x = ok[2 * e] ? xs[2 * e] * ws[2 * e] * ms[2 * e] : 0.f;
      v.y = ok[2 * e + 1] ? xs[2 * e + 1] * ws[2 * e + 1] * ms[2 * e + 1] : 0.f;
      data[fl * FS + i] = v;
    }
  }
  __syncthreads();
  fft_passes<LOGN, false>(data, tw, FB * NC);
  // split step + transposing epilogue: lanes along frames
  const int FO = d.frames_out;
  for (int idx = tid; idx < d.bins * FB; idx += 256) {
    const int k = idx / FB, fl = idx - k * FB;
    const int f = f_first + fl;
    if (f >= f_end) continue;
    const float2 A = data[fl * FS + rev[k & (NC - 1)]];
    const float2 Bq = data[fl * FS + rev[(NC - k) & (NC - 1)]];
    const float2 Bc = make_float2(Bq.x, -Bq.y);
    const float2 E = make_float2(0.5f * (A.x + Bc.x), 0.5f * (A.y + Bc.y));
    const float2 D = make_float2(0.5f * (A.x - Bc.x), 0.5f * (A.y - Bc.y));
    const float2 O = make_float2(D.y, -D.x);  // D * (-i)
    const float2 hw = half_twiddle<LOGN>(tw, tw2, k);      // e^{-i pi k / NC}
    float2 X = cmul(O, hw);
    X.x += E.x;
    X.y += E.y;
    if (d.herm) {  // gradient of irfft: middle bins doubled, DC / Nyquist imaginary part dropped
      if (k == 0 || k == NC) X.y = 0.f;
      else { X.x *= 2.f; X.y *= 2.f; }
    }
    const int fo = f - d.frame0;
    const int64_t rb = (int64_t)row * d.bins + k;
    switch (d.mode) {
      case RFX_STFT_COMPLEX:
        reinterpret_cast<float2*>(a.out)[rb * FO + fo] = X;
        break;
      case RFX_STFT_CAC:
        a.out[((int64_t)row * 2 * d.bins + k) * FO + fo] = X.x;
        a.out[((int64_t)row * 2 * d.bins + d.bins + k) * FO + fo] = X.y;
        break;
      case RFX_STFT_MAG:
        a.out[rb * FO + fo] = sqrtf(fmaxf(X.x * X.x + X.y * X.y, d.eps));
        break;
      case RFX_STFT_POW:
        a.out[rb * FO + fo] = X.x * X.x + X.y * X.y;
        break;
      default:  // RFX_STFT_MAGPOW
        a.out[rb * FO + fo] = powf(sqrtf(X.x * X.x + X.y * X.y) + d.eps, d.alpha);
        break;
    }
  }
}

template <int LOGN>
__global__ __launch_bounds__(256) void fft_synthesis_kernel(const FftArgs a) {
  constexpr int NC = 1 << LOGN, FB = RFX_FFT_PTS(LOGN) >> LOGN, FS = NC + 1, N = 2 * NC;
  __shared__ float2 data[FB * FS];
  __shared__ float2 tw[NC];
  __shared__ float2 tw2[LOGN <= 10 ? NC / 2 : 1];     // e^{-i pi k / NC} for ODD k (even k is tw[k/2]); LDS budget: n_fft <= 2048
  __shared__ uint16_t rev[NC];
  const rfx_stft_desc& d = a.d;
  int row, f_first;
  block_coords<LOGN>(a, row, f_first);
  if (row >= d.R) return;
  const int tid = threadIdx.x;
  build_twiddles<LOGN>(tw);
  build_half_twiddles<LOGN>(tw2);
  build_rev<LOGN>(rev);
  __syncthreads();                       // the merge step below reads the tables other threads built
  const int f_end = d.frame0 + d.frames_out;
  const int FO = d.frames_out;
  // merge step: Z[k] = (X[k] + conj X[NC-k]) + i e^{+i pi k/NC} (X[k] - conj X[NC-k]),  k in [0, NC)
  // batched, unconditional loads (see the analysis kernel): 2 points x (bin k, bin NC-k) x (re, im) per round
  for (int j0 = 0; j0 < (NC * FB) / 256; j0 += 2) {
    float2 xk[2], xm[2];
    bool fv[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int idx = tid + 256 * (j0 + e);
      const int k = idx / FB, fl = idx - k * FB;
      const int f = f_first + fl;
      fv[e] = f < f_end;
      const int fo = fv[e] ? f - d.frame0 : 0;
      auto fetch = [&](int kk) -> float2 {
        const bool in = kk < d.bins;
        const int kc = in ? kk : 0;
        float2 v;
        if (d.mode == RFX_STFT_COMPLEX) v = reinterpret_cast<const float2*>(a.x)[((int64_t)row * d.bins + kc) * FO + fo];   // uniform
        else {
          v.x = a.x[((int64_t)row * 2 * d.bins + kc) * FO + fo];
          v.y = a.x[((int64_t)row * 2 * d.bins + d.bins + kc) * FO + fo];
        }
        if (!in) v = make_float2(0.f, 0.f);
        if (kk == 0 || kk == NC) v.y = 0.f;           // real by construction / ignored by irfft
        else if (!d.herm) { v.x *= 0.5f; v.y *= 0.5f; }  // adjoint of the one-sided rfft
        return v;
      };
      xk[e] = fetch(k);
      xm[e] = fetch(NC - k);
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int idx = tid + 256 * (j0 + e);
      const int k = idx / FB, fl = idx - k * FB;
      float2 Z = make_float2(0.f, 0.f);
      if (fv[e]) {
        const float2 Xc = make_float2(xm[e].x, -xm[e].y);
        const float2 S = make_float2(xk[e].x + Xc.x, xk[e].y + Xc.y);
        const float2 D = make_float2(xk[e].x - Xc.x, xk[e].y - Xc.y);
        const float2 hw = half_twiddle<LOGN>(tw, tw2, k);     // e^{-i pi k/NC}; the merge needs its conjugate
        const float2 W = cmul(D, make_float2(hw.x, -hw.y));
        Z = make_float2(S.x - W.y, S.y + W.x);  // S + i*W
      }
      data[fl * FS + rev[k]] = Z;
    }
  }
  __syncthreads();
  fft_passes<LOGN, true>(data, tw, FB * NC);
  const int woff = (N - d.win) / 2;
  float* outr = a.out + (int64_t)row * d.T;
  // Overlap-add by GATHER inside the workgroup: its frames are consecutive, so every padded position p of their span
  // sums the <= ceil(win / hop) frames that cover it out of LDS and issues ONE global atomic (the scatter form issued
  // one per (frame, sample): 3.3-4.3x more, and the loss-gradient launches ran at the L2 atomic rate, 0.46 TB/s).
  const int nf = min(FB, f_end - f_first);
  if (nf <= 0) return;
  const int span = (nf - 1) * d.hop + N;
  const int64_t p0 = (int64_t)f_first * d.hop;
  for (int q = tid; q < span; q += 256) {
    const int qw = q - woff;                       // window index of frame 0 at this position
    if (qw < 0) continue;
    const int fl_hi = min(nf - 1, qw / d.hop);
    const int lo_num = qw - d.win + 1;
    const int fl_lo = lo_num > 0 ? (lo_num + d.hop - 1) / d.hop : 0;
    if (fl_lo > fl_hi) continue;
    float v = 0.f;
    for (int fl = fl_lo; fl <= fl_hi; ++fl) {
      const int t = q - fl * d.hop;
      const float2 z = data[fl * FS + (t >> 1)];
      v += ((t & 1) ? z.y : z.x) * a.window[t - woff];
    }
    const int64_t p = p0 + q;
    const int sidx = map_sample(d, (int)p);
    if (sidx < 0) continue;
    v *= d.scale;
    if (a.mul) v *= a.mul[p];
    atomicAdd(outr + sidx, v);
  }
}

static bool stft_desc_ok(const rfx_stft_desc* d) {
  if (!d) return false;
  const int n = d->n_fft;
  if (n != 512 && n != 1024 && n != 2048 && n != 4096) return false;
  return d->R > 0 && d->T > 0 && d->hop > 0 && d->win > 0 && d->win <= n && d->frames_out > 0 &&
         d->bins > 0 && d->bins <= n / 2 + 1 && d->frame0 >= 0;
}

template <bool SYN>
static int launch_fft(const rfx_stft_desc* d, const float* x, const float* window, const float* mul,
                      float* out, void* stream) {
  if (!stft_desc_ok(d) || !x || !window || !out) return -1;
  FftArgs a;
  a.d = *d; a.x = x; a.window = window; a.mul = mul; a.out = out;
  const int nc = d->n_fft / 2;
  const int fb = rfx_fft_pts(nc) / nc;
  a.groups_per_row = (d->frames_out + fb - 1) / fb;
  const int rows8 = (d->R + 7) / 8;
  const unsigned grid = (unsigned)(rows8 * a.groups_per_row * 8);
  hipStream_t s = (hipStream_t)stream;
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

extern "C" int rfx_fft_analysis(const rfx_stft_desc* d, const float* x, const float* window,
                                const float* mul, float* out, void* stream) {
  return launch_fft<false>(d, x, window, mul, out, stream);
}
extern "C" int rfx_fft_synthesis(const rfx_stft_desc* d, const float* spec, const float* window,
                                 const float* mul, float* out, void* stream) {
  if (d && d->mode != RFX_STFT_COMPLEX && d->mode != RFX_STFT_CAC) return -1;
  return launch_fft<true>(d, spec, window, mul, out, stream);
}
