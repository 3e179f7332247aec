// Gather-GEMM family, host entry points + the small kernels: weight packing, the thin (M <= 8) forward kernel.
// The tiled forward kernel lives in gemm_fwd.h (instantiated in gemm_fwd_f32.hip / gemm_fwd_bf3.hip), the weight
// gradient in gemm_wgrad.hip.
#include "gemm_fwd.h"

// ---------------------------------------------------------------------------------
// pack / unpack
// ---------------------------------------------------------------------------------
__global__ void pack_a_kernel(const float* __restrict__ w, const int32_t* __restrict__ woff,
                              int64_t w_ms, int M, int K, int Mpad, int Kpad,
                              float* __restrict__ apack) {
  const int64_t total = (int64_t)(Kpad + 64) * Mpad;   // four extra all-zero K steps (branch-free prefetch)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i / Mpad), m = (int)(i % Mpad);
    float v = 0.f;
    if (k < K && m < M) { const int o = woff[k]; v = o >= 0 ? w[(int64_t)m * w_ms + o] : 0.f; }
    apack[i] = v;
  }
}


// bf16x3 operand layout: two arrays (hi, lo) of [Kpad/8 + 8][Mpad] 16-byte cells, a cell = the 8
// consecutive-k bf16 values of one output row = exactly one lane's MFMA A fragment.
__global__ void pack_a_bf3_kernel(const float* __restrict__ w, const int32_t* __restrict__ woff,
                                  int64_t w_ms, int M, int K, int Mpad, int Kpad, uint4* __restrict__ apack) {
  const int64_t cells = (int64_t)(Kpad / 8 + 8) * Mpad;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int k8 = (int)(i / Mpad), m = (int)(i % Mpad);
    uint32_t hi[4] = {0, 0, 0, 0}, lo[4] = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int k = k8 * 8 + q;
      float v = 0.f;
      if (k < K && m < M) { const int o = woff[k]; v = o >= 0 ? w[(int64_t)m * w_ms + o] : 0.f; }
      const uint32_t h = bf16_rne(v);
      const uint32_t l = bf16_rne(v - __uint_as_float(h << 16));
      hi[q >> 1] |= h << (16 * (q & 1));
      lo[q >> 1] |= l << (16 * (q & 1));
    }
    apack[i] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    apack[cells + i] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
  }
}


// Thin forward kernel: M <= 8 output rows (TCN output conv 256->1, tcn.py:119,129;
// last HDemucs decoders).  HBM-bound: one thread per position, K loop with
// wave-uniform weights, coalesced gathers.
template <int MM>
__device__ __forceinline__ void thin_phase(const rfx_gemm_desc& d, const rfx_ktab_entry* __restrict__ kt,
                                           const float* __restrict__ ap, int K, const float* inb,
                                           const float* safe, bool jvalid, int ia0, int ib0,
                                           float (&acc)[MM]) {
  for (int k0 = 0; k0 < K; k0 += 4) {
    float bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const rfx_ktab_entry e = kt[k0 + u];  // Kpad is a multiple of 16: always readable
      const bool ok = jvalid && (unsigned)(ia0 + e.da) < (unsigned)d.IA &&
                      (unsigned)(ib0 + e.db) < (unsigned)d.IB;
      const float* p = ok ? inb + e.off : safe;
      const float v = *p;
      bv[u] = ok ? v : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int m = 0; m < MM; ++m) acc[m] = fmaf(ap[(int64_t)(k0 + u) * d.Mpad + m], bv[u], acc[m]);
  }
}

template <int MM>
__global__ __launch_bounds__(256) void gemm_thin_fwd_kernel(const FwdArgs g) {
  const rfx_gemm_desc& d = g.d;
  const int P = d.OA * d.OB;
  const int n = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const bool jvalid = j < P;
  const int jj = jvalid ? j : 0;
  const int a = jj / d.OB, b = jj - a * d.OB;
  const int ia0 = a * d.SA, ib0 = b * d.SB;
  const int64_t ioff = (int64_t)n * d.in_ns + (int64_t)ia0 * d.in_as + (int64_t)ib0 * d.in_bs;
  float acc[MM];
#pragma unroll
  for (int m = 0; m < MM; ++m) acc[m] = 0.f;
  thin_phase<MM>(d, g.ktab, g.apack, d.K, g.in + ioff, g.in, jvalid, ia0, ib0, acc);
  const rfx_epilogue& e = g.e;
#pragma unroll
  for (int m = 0; m < MM; ++m) {
    if (m < d.M) {
      float v = acc[m];
      if (e.bias) v += e.bias[m];
      if (e.act != RFX_ACT_NONE && !e.bwd)
        v = rfx_act_apply(v, e.act, e.act == RFX_ACT_PRELU ? e.act_param[m] : 0.f);
      acc[m] = v;
    }
  }
  if (g.apack2) {
    const float* in2 = g.in2 ? g.in2 : g.in;
    thin_phase<MM>(d, g.ktab2, g.apack2, g.Kpad2, in2 + ioff, in2, jvalid, ia0, ib0, acc);
  }
  if (!jvalid) return;
  float st1 = 0.f, st2 = 0.f;
  const int64_t opos = (int64_t)(a * d.out_sa + d.out_a0) * d.out_as + (int64_t)(b * d.out_sb + d.out_b0) * d.out_bs;
  float* outp = g.out + (int64_t)n * d.out_ns + opos;
#pragma unroll
  for (int m = 0; m < MM; ++m) {
    if (m < d.M) {
      float v = acc[m];
      float r = 0.f;
      if (e.res)
        r = e.res[(int64_t)n * e.res_ns + (int64_t)m * e.res_cs +
                  (int64_t)(a * d.out_sa + d.out_a0) * e.res_as + (int64_t)(b * d.out_sb + d.out_b0) * e.res_bs];
      if (e.bwd) {
        const float s = (e.act == RFX_ACT_PRELU) ? e.act_param[m] : 0.f;
        outp[(int64_t)m * d.out_cs] = r * rfx_act_grad(v, e.act, s);
        if (e.gparam && v < 0.f)
          atomicAdd(e.gparam + (int64_t)(e.stat_slots > 1 ? blockIdx.x & (e.stat_slots - 1) : 0) * d.M + m, r * v);
      } else {
        v += r;
        if (e.act2 != RFX_ACT_NONE) v = rfx_act_apply(v, e.act2, 0.f);
        outp[(int64_t)m * d.out_cs] = v;
        st1 += v; st2 += v * v;
      }
    }
  }
  if (e.stat_sums) {
    const int slots = e.stat_slots > 1 ? e.stat_slots : 1;     // thin path: tiny tensors, per-thread atomics are fine
    double* dst = e.stat_sums + 2 * ((int64_t)n * slots + (blockIdx.x & (slots - 1)));
    atomicAdd(dst, (double)st1);
    atomicAdd(dst + 1, (double)st2);
  }
}


extern "C" int rfx_abi_version(void) { return RFX_ABI_VERSION; }

extern "C" int rfx_pack_a(const float* w, const int32_t* woff, int64_t w_ms, int32_t M, int32_t K,
                          int32_t Mpad, int32_t Kpad, int32_t prec, float* apack, void* stream) {
  if (!w || !woff || !apack || M <= 0 || K < 0 || Mpad < M || Kpad < K) return -1;
  if (prec != 0) {
    const int64_t cells = (int64_t)(Kpad / 8 + 8) * Mpad;
    const int grid = (int)((cells + 255) / 256 < 4096 ? (cells + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_a_bf3_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, woff, w_ms, M, K, Mpad,
                       Kpad, reinterpret_cast<uint4*>(apack));
    RFX_CHECK_LAUNCH();
    return 0;
  }
  const int64_t total = (int64_t)(Kpad + 64) * Mpad;
  if (total == 0) return 0;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_a_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, woff, w_ms, M, K,
                     Mpad, Kpad, apack);
  RFX_CHECK_LAUNCH();
  return 0;
}


static int pick_r(int M, int K) {
  if (M <= 8) return 0;  // thin path
  if (M <= 32) return 1;
  // short reductions are output-write bound: one 32-row tile per wave keeps the kernel at ~100 VGPRs
  // (4 waves per SIMD instead of 2) so more stores / gathers are in flight per CU
  if (K <= 64) return 1;
  int best = 4, best_pad = ((M + 127) / 128) * 128;
  for (int r = 3; r >= 2; --r) {
    const int bm = 32 * r, pad = ((M + bm - 1) / bm) * bm;
    if (pad < best_pad) { best = r; best_pad = pad; }
  }
  return best;
}

extern "C" int rfx_gemm_pick_r(int32_t M, int32_t K) { return pick_r(M, K); }

extern "C" int rfx_gemm_fwd(const rfx_gemm_desc* d, const float* apack, const rfx_ktab_entry* ktab,
                            const float* in, float* out, const rfx_epilogue* epi, const float* apack2,
                            const rfx_ktab_entry* ktab2, int32_t K2, int32_t Kpad2, const float* in2,
                            int32_t prec, void* stream) {
  if (!desc_ok(d) || !apack || !ktab || !in || !out) return -1;
  if ((apack2 != nullptr) != (ktab2 != nullptr)) return -1;
  if (apack2 && (Kpad2 % 16 != 0 || (prec == 0 && Kpad2 < K2))) return -1;
  if (prec < 0 || prec > 2 || (d->R == 0 && prec != 0)) return -1;
  FwdArgs g;
  g.xcd_chunk = 0;
  g.pair_store = 1;
  g.d = *d;
  g.apack = apack; g.ktab = ktab; g.in = in; g.out = out;
  if (epi) g.e = *epi;
  else { g.e = rfx_epilogue{}; }
  if (g.e.bwd && !g.e.res) return -1;
  if (g.e.glu_out && (d->R == 0 || (d->M & 1) || g.e.bwd || g.e.res || g.e.act2 != RFX_ACT_NONE || g.e.stat_sums || apack2 ||
                      d->mg_log)) return -1;
  if (d->mg_log && (d->R == 0 || g.e.bwd || g.e.act2 != RFX_ACT_NONE || g.e.stat_sums || apack2 ||
                    d->mg_log > 8 || d->mg_axis < 0 || d->mg_axis > 1)) return -1;
  // bf16 storage of single operands: gathered input on the tap-major tiled kernels of the bf16 mode; output wherever the
  // plain / GLU store runs (not the phase-merged, backward-epilogue or thin paths)
  if (d->in_bf16 && (prec != 2 || d->R == 0)) return -1;
  if (d->in_bf16 == 2) return -1;                      // (was: paired gather of 1x1 plans, removed)
  if (d->out_bf16 && (d->R == 0 || d->mg_log || g.e.bwd || g.e.res)) return -1;
  g.apack2 = apack2; g.ktab2 = ktab2; g.Kpad2 = apack2 ? Kpad2 : 0; g.in2 = in2;
  g.ntaps2 = apack2 ? K2 : 0;      // tap-major launches pass the second phase's tap count in K2
  const int P = d->OA * d->OB;
  const int r = d->R;
  if (r < 0 || r > 4 || (r == 0 && d->M > 8)) return -1;       // M <= 8 is normally thin (R = 0); merged-phase plans ask for R = 1
  hipStream_t s = (hipStream_t)stream;
  if (r == 0) {
    dim3 grid((P + 255) / 256, d->N);
    if (d->M <= 1) hipLaunchKernelGGL(gemm_thin_fwd_kernel<1>, grid, dim3(256), 0, s, g);
    else if (d->M <= 2) hipLaunchKernelGGL(gemm_thin_fwd_kernel<2>, grid, dim3(256), 0, s, g);
    else if (d->M <= 4) hipLaunchKernelGGL(gemm_thin_fwd_kernel<4>, grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL(gemm_thin_fwd_kernel<8>, grid, dim3(256), 0, s, g);
    RFX_CHECK_LAUNCH();
    return 0;
  }
  // stride-1 multi-tap plans in the bf16 mode: halo-tile kernel (gemm_halo.h) -- the input tile of a 256-position row segment is
  // staged once per 16-channel chunk in LDS and serves every tap
  // (the halo kernel builds 32-bit byte offsets against a sample descriptor too: same extent limits as the tap-major path below)
  if (prec == 2 && !apack2 && d->in_bf16 != 3 && rfx_halo_takes(*d)) {
    if (d->in_extent <= 0 || d->in_extent > 0x7fffffffLL) return -1;
    return rfx_launch_gemm_halo(g, s);
  }
  const int bm = 32 * r;
  if (d->Mpad % bm != 0) return -1;
  const int64_t work = (int64_t)((P + 127) / 128) * d->N;          // (sample, position tile) items
  const int64_t nblk = ((work + 7) / 8) * 8 * (d->Mpad / bm);
  if (nblk > 0x7fffffff) return -1;
  dim3 grid((unsigned)nblk);
  // Position tiles are 128 flattened (a, b) positions.  When the taps reach over rows of the A axis (2-D kernels, the (8,1) / stride-4
  // frequency convolutions and their transposes) neighbouring tiles read the same input rows: give each XCD a contiguous run of
  // tiles so the overlap is served by ITS L2 instead of being fetched from HBM once per XCD.
  g.xcd_chunk = d->OA > 1 ? (int)((work + 7) / 8) : 0;
  if (prec == 0) return rfx_launch_gemm_fwd_f32(g, r, grid, s);
  // bf16x3 / bf16: tap-major kernels (gemm_tap.h); the caller packed A and passes the tap tables in that order
  if (d->Kpad_t <= 0 || d->Kpad_t % 16 != 0 || d->gpt <= 0 || d->ntaps <= 0 || d->ntaps > 112 || d->in_extent <= 0 ||
      d->in_extent > 0x7fffffffLL || d->in_cs * 4 * 8 * d->gpt > 0x7fffffffLL) return -1;
  if (apack2 && (K2 <= 0 || K2 > 112)) return -1;
  return prec == 1 ? rfx_launch_gemm_fwd_bf3(g, r, grid, s) : rfx_launch_gemm_fwd_bf16(g, r, grid, s);
}

// Which kernel instantiation rfx_gemm_fwd launches for (desc, epilogue, prec): measurement tools label launches with it
// (bench.py's per-kernel roofline).  kind * 16 + R; kind: 0 gemm_thin_fwd_kernel<M>, 1 gemm_fwd_kernel<R> (exact fp32, channel-major),
// 2 gemm_tap_kernel<R, prec>, 3 gemm_tap_kernel<R, 2, IN16> (16-bit gathered operand), 4 gemm_tap_stream_kernel<prec, 4, 1>,
// 5 gemm_tap_stream_kernel<prec, 2, 4>, 6 / 7 gemm_halo_kernel<R, 9, 0 / 1> (fp32 / 16-bit operand), 8 / 9 gemm_halo_kernel<R, 3, 0 / 1>.
// Pure function of its arguments.
extern "C" int rfx_gemm_fwd_variant(const rfx_gemm_desc* d, const rfx_epilogue* epi, int32_t two_phase, int32_t prec) {
  if (!d || prec < 0 || prec > 2) return -1;
  rfx_epilogue e = epi ? *epi : rfx_epilogue{};
  const int r = d->R;
  if (r == 0) return 0;
  if (prec == 0) return 16 + r;
  if (prec == 2 && !two_phase && d->in_bf16 != 3 && rfx_halo_takes(*d))
    return 16 * (6 + (d->in_bf16 ? 1 : 0) + (d->halo_nt == 3 ? 2 : 0)) + rfx_halo_pick_r(*d);
  if (d->in_bf16) return 48 + r;
  if (rfx_tap_use_stream(*d, e, two_phase != 0, r)) return (d->Kpad_t <= 16 ? 64 : 80) + r;
  return 32 + r;
}

