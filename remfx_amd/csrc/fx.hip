// Audio-effect rendering on the device: the five effects of remfx/effects.py:297-616 that the
// reference renders on the CPU with pedalboard (JUCE DSP) while it builds / augments the dataset, and the BS.1770
// loudness normalisation it applies after every effect (effects.py:619-629, pyloudnorm).  All kernels take a batch of
// mono clips x: (B, T) fp32 contiguous and PER-CLIP parameter vectors (every clip draws its own random parameters,
// datasets.py:109-202 / 205-330), so one launch renders a whole training batch.
//
// These are HBM-bound byte-streaming or latency-bound recurrence kernels, not GEMMs: coalesced reads along T, recurrences
// blocked by their own lag (a delay line of D samples makes D consecutive outputs independent), wave-level scans where
// the lag is one sample.  Algorithms restated from the published JUCE / pedalboard / pyloudnorm sources (absent from
// the image: parity unpinned, oracle/ref_effects.py is the same restatement in numpy float64).
#include "common.h"

// ---- distortion: pedalboard.Distortion = JUCE Gain(drive_db) -> WaveShaper(tanh) ------------------------------------
__global__ __launch_bounds__(256) void fx_distortion_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t T,
                                                            const float* __restrict__ gain) {
  const int b = blockIdx.y;
  const float g = gain[b];
  const float* xr = x + (int64_t)b * T;
  float* yr = y + (int64_t)b * T;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < T; i += (int64_t)gridDim.x * 1024) {
    if (i + 3 < T) {
      f32x4 v = rfx_ld4(xr + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = tanhf(g * v[j]);
      rfx_st4(yr + i, v);
    } else {
      for (int64_t k = i; k < T; ++k) yr[k] = tanhf(g * xr[k]);
    }
  }
}

// ---- delay: pedalboard.Delay = JUCE DelayLine (integer delay D = int(seconds * sr)), feedback, dry/wet mix ------------
//   delayed[n] = w[n - D];  w[n] = x[n] + fb * delayed[n];  y[n] = (1 - mix) x[n] + mix * delayed[n]
// Unrolled in closed form (the line starts empty): delayed[n] = sum_{k >= 1} fb^(k-1) x[n - k D]: every output sample is
// independent, no workspace, <= T / D taps (D >= 0.1 s in the reference's ranges).
__global__ __launch_bounds__(256) void fx_delay_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t T,
                                                       const int32_t* __restrict__ delay, const float* __restrict__ fb,
                                                       const float* __restrict__ mix) {
  const int b = blockIdx.y;
  const int64_t D = delay[b];
  const float f = fb[b], m = mix[b];
  const float* xr = x + (int64_t)b * T;
  float* yr = y + (int64_t)b * T;
  for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < T; n += (int64_t)gridDim.x * 256) {
    float acc = 0.f, w = 1.f;
    if (D > 0)
      for (int64_t k = n - D; k >= 0; k -= D) { acc = fmaf(w, xr[k], acc); w *= f; }
    else acc = 0.f;
    yr[n] = (1.0f - m) * xr[n] + m * acc;
  }
}

// ---- chorus: JUCE dsp::Chorus (pedalboard.Chorus) -----------------------------------------------------------------------
//   lfo[n] = sin(2 pi rate n / sr - pi) * depth / 2;   d[n] = max(1 ms, 20 ms * lfo[n] + centre) * sr / 1000  (samples)
//   pushed[n] = x[n] - fb * popped[n - 1];  popped[n] = linear interpolation of `pushed` at n - d[n];  y = (1 - mix) x + mix * popped
// The feedback lags ONE sample but reaches back at least 1 ms (48 samples at 48 kHz): blocks of BLK = 32 consecutive
// samples are independent given the history.  One wave per clip, the `pushed` history in an LDS ring.
#define FX_CH_RING 4096
__global__ __launch_bounds__(64) void fx_chorus_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t T, float sr,
                                                       const float* __restrict__ rate, const float* __restrict__ depth,
                                                       const float* __restrict__ centre_ms, const float* __restrict__ fb,
                                                       const float* __restrict__ mix, int blk) {
  __shared__ float ring[FX_CH_RING];
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* xr = x + (int64_t)b * T;
  float* yr = y + (int64_t)b * T;
  const float f = fb[b], m = mix[b], dep = 0.5f * depth[b], cen = centre_ms[b];
  const double winc = 2.0 * 3.14159265358979323846 * (double)rate[b] / (double)sr;
  for (int i = lane; i < FX_CH_RING; i += 64) ring[i] = 0.f;
  __syncthreads();
  float prev_pop = 0.f;                       // popped[n0 - 1]
  for (int64_t n0 = 0; n0 < T; n0 += blk) {
    const int64_t n = n0 + lane;
    const bool act = lane < blk && n < T;
    float pop = 0.f, xv = 0.f;
    if (act) {
      xv = xr[n];
      double ph = winc * (double)n;                     // sin(ph - pi) = -sin(ph); reduce in fp64, evaluate in fp32
      ph -= floor(ph * 0.15915494309189535) * 6.283185307179586;
      const float lfo = -sinf((float)ph) * dep;
      const float dms = fmaxf(1.0f, 20.0f * lfo + cen);
      const float d = dms * sr / 1000.0f;
      const int di = (int)d;
      const float fr = d - (float)di;
      const int64_t i1 = n - di, i2 = i1 - 1;
      const float v1 = i1 >= 0 ? ring[i1 & (FX_CH_RING - 1)] : 0.f;
      const float v2 = i2 >= 0 ? ring[i2 & (FX_CH_RING - 1)] : 0.f;
      pop = v1 + fr * (v2 - v1);
    }
    // pushed[n] = x[n] - fb * popped[n - 1]: previous lane's pop (lane 0: carried from the last block)
    float pm1 = __shfl_up(pop, 1, 64);
    if (lane == 0) pm1 = prev_pop;
    prev_pop = __shfl(pop, blk - 1, 64);
    __syncthreads();                          // all ring reads of this block are done
    if (act) {
      ring[n & (FX_CH_RING - 1)] = xv - f * pm1;
      yr[n] = (1.0f - m) * xv + m * pop;
    }
    __syncthreads();
  }
}

// ---- compressor: JUCE dsp::Compressor (pedalboard.Compressor): peak ballistics envelope + static gain computer ---------
//   env[n] = a + c (env[n-1] - a),  a = |x[n]|, c = (a > env[n-1]) ? c_attack : c_release,   c_* = exp(-2 pi 1000 / (sr ms))
//   gain = env < thr ? 1 : (env / thr)^(1 / ratio - 1);   y = gain * x
// A nonlinear one-sample recurrence: one LANE per clip walks its clip sequentially (the envelope only), then the gain is
// applied by all lanes.  64 clips = one wave; the envelope is written to a workspace so the second pass is coalesced.
__global__ __launch_bounds__(64) void fx_comp_env_kernel(const float* __restrict__ x, float* __restrict__ env, int B, int64_t T,
                                                         const float* __restrict__ c_at, const float* __restrict__ c_rl) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const float ca = c_at[b], cr = c_rl[b];
  const float* xr = x + (int64_t)b * T;
  float* er = env + (int64_t)b * T;
  float yv = 0.f;
  // the recurrence is a ~20-cycle dependent chain per sample; the loads do not depend on it: 8 x 16 bytes in flight per lane
  int64_t n = 0;
  for (; n + 32 <= T; n += 32) {
    f32x4 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = rfx_ld4(xr + n + 4 * q);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = fabsf(v[q][j]);
        const float c = a > yv ? ca : cr;
        yv = a + c * (yv - a);
        o[j] = yv;
      }
      rfx_st4(er + n + 4 * q, o);
    }
  }
  for (; n < T; ++n) {
    const float a = fabsf(xr[n]);
    const float c = a > yv ? ca : cr;
    yv = a + c * (yv - a);
    er[n] = yv;
  }
}
__global__ __launch_bounds__(256) void fx_comp_gain_kernel(const float* __restrict__ x, const float* __restrict__ env,
                                                           float* __restrict__ y, int64_t T, const float* __restrict__ thr,
                                                           const float* __restrict__ ratio) {
  const int b = blockIdx.y;
  const float th = thr[b], ti = 1.0f / th, ex = 1.0f / ratio[b] - 1.0f;
  const float* xr = x + (int64_t)b * T;
  const float* er = env + (int64_t)b * T;
  float* yr = y + (int64_t)b * T;
  for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < T; n += (int64_t)gridDim.x * 256) {
    const float e = er[n];
    const float g = e < th ? 1.0f : powf(e * ti, ex);
    yr[n] = g * xr[n];
  }
}

// ---- reverb: JUCE Reverb (Freeverb; pedalboard.Reverb), mono path ----------------------------------------------------
//   in = 0.015 x;  out = sum_j comb_j(in);  out = allpass_3(allpass_2(allpass_1(allpass_0(out))));  y = wet1 * out + dry * x
//   comb:    o = buf[i]; last = o (1 - damp) + last damp; buf[i] = in + last * feedback; return o          (lag = comb length)
//   allpass: o = buf[i]; buf[i] = in + 0.5 o; return o - in                                                (lag = length)
// Every filter's lag (>= 225 * sr / 44100 samples) exceeds a 64-sample block, so one wave per clip renders 64 samples per
// iteration; the one-pole damping filter inside a comb is a one-sample linear recurrence = a 6-step wave scan.
// All 12 delay buffers of a clip live in LDS.
#define FX_RV_NC 8
#define FX_RV_NA 4
struct FxReverbArgs {
  const float* x;
  float* y;
  int64_t T;
  const float *damp, *feedback, *wet1, *dry;     // per clip
  int comb_len[FX_RV_NC], ap_len[FX_RV_NA];
  int comb_off[FX_RV_NC], ap_off[FX_RV_NA];      // offsets in the LDS arena
  int arena;
};
__global__ __launch_bounds__(64) void fx_reverb_kernel(const FxReverbArgs a) {
  extern __shared__ float arena[];
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* xr = a.x + (int64_t)b * a.T;
  float* yr = a.y + (int64_t)b * a.T;
  const float damp = a.damp[b], fbk = a.feedback[b], wet1 = a.wet1[b], dry = a.dry[b];
  for (int i = lane; i < a.arena; i += 64) arena[i] = 0.f;
  float last[FX_RV_NC];
  int cpos[FX_RV_NC], apos[FX_RV_NA];
#pragma unroll
  for (int j = 0; j < FX_RV_NC; ++j) { last[j] = 0.f; cpos[j] = 0; }
#pragma unroll
  for (int j = 0; j < FX_RV_NA; ++j) apos[j] = 0;
  // damp^(2^s) for the scan, damp^(lane + 1) for the carried state
  float dpw[6];
  dpw[0] = damp;
#pragma unroll
  for (int s = 1; s < 6; ++s) dpw[s] = dpw[s - 1] * dpw[s - 1];
  const float dl1 = powf(damp, (float)(lane + 1));
  __syncthreads();
  for (int64_t n0 = 0; n0 < a.T; n0 += 64) {
    const int64_t n = n0 + lane;
    const bool act = n < a.T;
    const float xv = act ? xr[n] : 0.f;
    const float in = xv * 0.015f;
    float out = 0.f;
#pragma unroll
    for (int j = 0; j < FX_RV_NC; ++j) {
      float* buf = arena + a.comb_off[j];
      int idx = cpos[j] + lane;
      idx -= idx >= a.comb_len[j] ? a.comb_len[j] : 0;
      const float o = buf[idx];
      // last[i] = (1 - damp) o[i] + damp last[i - 1]: inclusive weighted scan over the 64 lanes
      float v = (1.0f - damp) * o;
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        const float u = __shfl_up(v, 1 << s, 64);
        if (lane >= (1 << s)) v = fmaf(dpw[s], u, v);
      }
      v = fmaf(dl1, last[j], v);
      buf[idx] = in + v * fbk;               // lanes beyond T write garbage-free values (in = 0) that are never read back in range
      last[j] = __shfl(v, 63, 64);
      cpos[j] += 64;
      cpos[j] -= cpos[j] >= a.comb_len[j] ? a.comb_len[j] : 0;
      out += o;
    }
#pragma unroll
    for (int j = 0; j < FX_RV_NA; ++j) {
      float* buf = arena + a.ap_off[j];
      int idx = apos[j] + lane;
      idx -= idx >= a.ap_len[j] ? a.ap_len[j] : 0;
      const float o = buf[idx];
      buf[idx] = out + 0.5f * o;
      out = o - out;
      apos[j] += 64;
      apos[j] -= apos[j] >= a.ap_len[j] ? a.ap_len[j] : 0;
    }
    if (act) yr[n] = out * wet1 + xv * dry;
  }
}

// ---- BS.1770 integrated loudness (pyloudnorm.Meter.integrated_loudness) + gain --------------------------------------
// K-weighting = two biquads in series evaluated in fp64 (scipy.signal.lfilter on float64).  A 4th-order LINEAR recurrence:
// the clip is cut into 64 chunks, one lane each.  Pass 1: every lane filters its chunk from a ZERO state and keeps the
// final state; the true initial states follow from s_k = M s_(k-1) + z_k with M = (state transition)^chunk (4 x 4, from the
// host); pass 2 re-filters with the right initial state and accumulates the squared output into 100 ms hop sums.
// The gating (400 ms blocks, 75 % overlap, -70 LUFS absolute and -10 LU relative gates) runs on the ~55 hop sums.
struct FxLoudArgs {
  const float* x;
  double* hop;            // (B, nhop) sums of squares of the K-weighted signal per hop (zeroed by the caller)
  int64_t T;
  int chunk, nhop;
  const int32_t* hop_of;  // not used: hops are uniform (hop_len) except that block bounds come from blk_lo / blk_hi
  int hop_len;
  double b1[3], a1[3], b2[3], a2[3];     // normalised (a[0] = 1)
  double M[16];                          // state transition of `chunk` samples, row-major 4 x 4 (states: s1a, s1b, s2a, s2b)
};
// transposed direct form II (scipy lfilter): y = b0 x + s0; s0 = b1 x - a1 y + s1; s1 = b2 x - a2 y
__device__ __forceinline__ double fx_biquad(const double* b, const double* a, double x, double& s0, double& s1) {
  const double y = b[0] * x + s0;
  s0 = b[1] * x - a[1] * y + s1;
  s1 = b[2] * x - a[2] * y;
  return y;
}
__global__ __launch_bounds__(64) void fx_kweight_kernel(const FxLoudArgs a) {
  __shared__ double st[64][4];
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* xr = a.x + (int64_t)b * a.T;
  const int64_t lo = (int64_t)lane * a.chunk, hi = lo + a.chunk < a.T ? lo + a.chunk : a.T;
  double s[4] = {0, 0, 0, 0};
  for (int64_t n = lo; n < hi; ++n) {
    const double v = fx_biquad(a.b1, a.a1, (double)xr[n], s[0], s[1]);
    fx_biquad(a.b2, a.a2, v, s[2], s[3]);
  }
  // chunks shorter than `chunk` (the tail) would need their own transition; only the LAST non-empty chunk can be short and
  // its end state is never used
#pragma unroll
  for (int i = 0; i < 4; ++i) st[lane][i] = s[i];
  __syncthreads();
  if (lane == 0) {                       // 64 tiny 4x4 steps: true state at the START of every chunk
    double cur[4] = {0, 0, 0, 0};
    for (int k = 0; k < 64; ++k) {
      double z[4], nxt[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { z[i] = st[k][i]; st[k][i] = cur[i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        double acc = z[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += a.M[i * 4 + j] * cur[j];
        nxt[i] = acc;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) cur[i] = nxt[i];
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) s[i] = st[lane][i];
  double acc = 0.0;
  int64_t h = lo / a.hop_len;
  int64_t hend = (h + 1) * a.hop_len;
  double* hp = a.hop + (int64_t)b * a.nhop;
  for (int64_t n = lo; n < hi; ++n) {
    if (n == hend) {
      if (h < a.nhop) atomicAdd(hp + h, acc);
      acc = 0.0; ++h; hend += a.hop_len;
    }
    const double v = fx_biquad(a.b1, a.a1, (double)xr[n], s[0], s[1]);
    const double w = fx_biquad(a.b2, a.a2, v, s[2], s[3]);
    acc += w * w;
  }
  if (hi > lo && h < a.nhop) atomicAdd(hp + h, acc);
}
// gating on the hop sums; gain[b] = 10^(clamp(target - L, -120, 40) / 20).  One thread per clip.
__global__ void fx_loud_gate_kernel(const double* __restrict__ hop, int B, int nhop, int nblk, double inv_block, float target,
                                    float* __restrict__ lufs, float* __restrict__ gain) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double* hp = hop + (int64_t)b * nhop;
  // block j = hops j .. j + 3 (400 ms at 75 % overlap)
  double s1 = 0.0; int n1 = 0;
  for (int j = 0; j < nblk; ++j) {
    const double z = (hp[j] + hp[j + 1] + hp[j + 2] + hp[j + 3]) * inv_block;
    const double l = -0.691 + 10.0 * log10(z);
    if (l >= -70.0) { s1 += z; ++n1; }
  }
  double L;
  if (n1 == 0) L = -INFINITY;
  else {
    const double gamma_r = -0.691 + 10.0 * log10(s1 / n1) - 10.0;
    double s2 = 0.0; int n2 = 0;
    for (int j = 0; j < nblk; ++j) {
      const double z = (hp[j] + hp[j + 1] + hp[j + 2] + hp[j + 3]) * inv_block;
      const double l = -0.691 + 10.0 * log10(z);
      if (l > gamma_r && l > -70.0) { s2 += z; ++n2; }
    }
    L = n2 ? -0.691 + 10.0 * log10(s2 / n2) : -INFINITY;
  }
  lufs[b] = (float)L;
  float d = target - (float)L;
  d = fminf(fmaxf(d, -120.0f), 40.0f);
  gain[b] = powf(10.0f, d / 20.0f);
}
__global__ __launch_bounds__(256) void fx_scale_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t T,
                                                       const float* __restrict__ gain) {
  const int b = blockIdx.y;
  const float g = gain[b];
  const float* xr = x + (int64_t)b * T;
  float* yr = y + (int64_t)b * T;
  for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < T; n += (int64_t)gridDim.x * 256) yr[n] = g * xr[n];
}

// ---- C ABI -----------------------------------------------------------------------------------------------------------------
static dim3 fx_grid(int64_t T, int B, int per_block) {
  int64_t gx = (T + per_block - 1) / per_block;
  if (gx > 2048) gx = 2048;
  if (gx < 1) gx = 1;
  return dim3((unsigned)gx, (unsigned)B);
}
static bool fx_ok(const void* x, const void* y, int B, int64_t T) { return x && y && B > 0 && B <= 65535 && T > 0; }

extern "C" int rfx_fx_distortion(const float* x, float* y, int32_t B, int64_t T, const float* gain, void* stream) {
  if (!fx_ok(x, y, B, T) || !gain) return -1;
  hipLaunchKernelGGL(fx_distortion_kernel, fx_grid(T, B, 1024), dim3(256), 0, (hipStream_t)stream, x, y, T, gain);
  RFX_CHECK_LAUNCH();
  return 0;
}
extern "C" int rfx_fx_delay(const float* x, float* y, int32_t B, int64_t T, const int32_t* delay_samples, const float* feedback,
                            const float* mix, void* stream) {
  if (!fx_ok(x, y, B, T) || !delay_samples || !feedback || !mix || x == y) return -1;
  hipLaunchKernelGGL(fx_delay_kernel, fx_grid(T, B, 256), dim3(256), 0, (hipStream_t)stream, x, y, T, delay_samples, feedback, mix);
  RFX_CHECK_LAUNCH();
  return 0;
}
extern "C" int rfx_fx_chorus(const float* x, float* y, int32_t B, int64_t T, float sample_rate, const float* rate_hz,
                             const float* depth, const float* centre_delay_ms, const float* feedback, const float* mix,
                             void* stream) {
  if (!fx_ok(x, y, B, T) || !rate_hz || !depth || !centre_delay_ms || !feedback || !mix || sample_rate < 4000.f) return -1;
  // blocks no longer than the 1 ms floor of the modulated delay minus the interpolation tap; the ring must hold the longest
  // delay the reference's ranges allow (centre + 20 ms * depth / 2, plus a block)
  int blk = (int)(sample_rate / 1000.0f) - 2;
  blk = blk > 64 ? 64 : blk;
  if (blk < 1 || (int)(sample_rate * 0.05f) + 66 > FX_CH_RING) return -1;
  hipLaunchKernelGGL(fx_chorus_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, x, y, T, sample_rate, rate_hz, depth,
                     centre_delay_ms, feedback, mix, blk);
  RFX_CHECK_LAUNCH();
  return 0;
}
extern "C" int rfx_fx_compressor(const float* x, float* y, float* env_ws, int32_t B, int64_t T, const float* threshold_lin,
                                 const float* ratio, const float* c_attack, const float* c_release, void* stream) {
  if (!fx_ok(x, y, B, T) || !env_ws || !threshold_lin || !ratio || !c_attack || !c_release) return -1;
  hipLaunchKernelGGL(fx_comp_env_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, x, env_ws, B, T, c_attack, c_release);
  hipLaunchKernelGGL(fx_comp_gain_kernel, fx_grid(T, B, 256), dim3(256), 0, (hipStream_t)stream, x, env_ws, y, T, threshold_lin, ratio);
  RFX_CHECK_LAUNCH();
  return 0;
}
extern "C" int rfx_fx_reverb(const float* x, float* y, int32_t B, int64_t T, int32_t sample_rate, const float* damp,
                             const float* feedback, const float* wet1, const float* dry, void* stream) {
  if (!fx_ok(x, y, B, T) || !damp || !feedback || !wet1 || !dry || sample_rate < 8000 || sample_rate > 192000) return -1;
  static const int comb_t[FX_RV_NC] = {1116, 1188, 1277, 1356, 1422, 1491, 1557, 1617};
  static const int ap_t[FX_RV_NA] = {556, 441, 341, 225};
  FxReverbArgs a{};
  a.x = x; a.y = y; a.T = T; a.damp = damp; a.feedback = feedback; a.wet1 = wet1; a.dry = dry;
  int off = 0;
  for (int j = 0; j < FX_RV_NC; ++j) {
    a.comb_len[j] = (int)(((int64_t)sample_rate * comb_t[j]) / 44100);
    if (a.comb_len[j] < 64) return -1;
    a.comb_off[j] = off; off += a.comb_len[j];
  }
  for (int j = 0; j < FX_RV_NA; ++j) {
    a.ap_len[j] = (int)(((int64_t)sample_rate * ap_t[j]) / 44100);
    if (a.ap_len[j] < 64) return -1;
    a.ap_off[j] = off; off += a.ap_len[j];
  }
  a.arena = off;
  const size_t lds = sizeof(float) * (size_t)off;
  if (lds > 160 * 1024) return -1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(fx_reverb_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
      hipSuccess) return -3;
  hipLaunchKernelGGL(fx_reverb_kernel, dim3(B), dim3(64), lds, (hipStream_t)stream, a);
  RFX_CHECK_LAUNCH();
  return 0;
}
extern "C" int rfx_fx_loudness(const float* x, int32_t B, int64_t T, int32_t chunk, int32_t hop_len, int32_t nhop, int32_t nblk,
                               double inv_block, const double* coef /* b1[3] a1[3] b2[3] a2[3] M[16] on the HOST */,
                               float target_lufs, double* hop_ws, float* lufs, float* gain, void* stream) {
  if (!x || !coef || !hop_ws || !lufs || !gain || B <= 0 || T <= 0 || chunk <= 0 || hop_len <= 0 || nhop < 4 || nblk < 1 ||
      nblk + 3 > nhop || (int64_t)chunk * 64 < T) return -1;
  FxLoudArgs a{};
  a.x = x; a.hop = hop_ws; a.T = T; a.chunk = chunk; a.nhop = nhop; a.hop_len = hop_len; a.hop_of = nullptr;
  for (int i = 0; i < 3; ++i) { a.b1[i] = coef[i]; a.a1[i] = coef[3 + i]; a.b2[i] = coef[6 + i]; a.a2[i] = coef[9 + i]; }
  for (int i = 0; i < 16; ++i) a.M[i] = coef[12 + i];
  if (hipMemsetAsync(hop_ws, 0, sizeof(double) * (size_t)B * nhop, (hipStream_t)stream) != hipSuccess) return -3;
  hipLaunchKernelGGL(fx_kweight_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(fx_loud_gate_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, hop_ws, B, nhop, nblk, inv_block,
                     target_lufs, lufs, gain);
  RFX_CHECK_LAUNCH();
  return 0;
}
extern "C" int rfx_fx_scale(const float* x, float* y, int32_t B, int64_t T, const float* gain, void* stream) {
  if (!fx_ok(x, y, B, T) || !gain) return -1;
  hipLaunchKernelGGL(fx_scale_kernel, fx_grid(T, B, 256), dim3(256), 0, (hipStream_t)stream, x, y, T, gain);
  RFX_CHECK_LAUNCH();
  return 0;
}
