// Weight-gradient kernels of the gather-GEMM family (see gemm_fwd.h for the family overview).
#include "gemm_wgrad.h"

// dw (+)= sum over the position splits' slices of dapack, in split order (fixed tree: 4 groups of consecutive splits per cell,
// each added in order with 8 loads in flight, then the 4 group sums in order) -- the weight gradients carry no atomics.
template <bool ADD>
__global__ __launch_bounds__(256) void unpack_add_kernel(const float* __restrict__ dapack, const int32_t* __restrict__ woff,
                                                         int64_t w_ms, int M, int K, int Kpad, float* __restrict__ dw, int splits,
                                                         int64_t split_stride, int bias_col = -1, float* __restrict__ db = nullptr) {
  __shared__ float part[4][64];
  __shared__ float bpart[4][64];
  const int l = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const int64_t total = (int64_t)K * M;
  const int per = (splits + 3) / 4;
  const int s0 = sg * per, s1 = min(s0 + per, splits);
  for (int64_t base = (int64_t)blockIdx.x * 64; base < total; base += (int64_t)gridDim.x * 64) {
    const int64_t i = base + l;
    const bool ok = i < total;
    const int m = ok ? (int)(i / K) : 0, k = ok ? (int)(i % K) : 0;
    const bool isb = ok && db != nullptr && k == 0;              // this thread also carries row m's bias-gradient column
    float sum = 0.f, bsum = 0.f;
    if (ok) {
      const float* src = dapack + (int64_t)m * Kpad + k;
      int s = s0;
      for (; s + 8 <= s1; s += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(s + u) * split_stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) sum += v[u];
      }
      for (; s < s1; ++s) sum += src[(int64_t)s * split_stride];
      if (isb)
        for (int q = s0; q < s1; ++q) bsum += dapack[(int64_t)q * split_stride + (int64_t)m * Kpad + bias_col];
    }
    part[sg][l] = sum;
    bpart[sg][l] = bsum;
    __syncthreads();
    if (sg == 0 && ok) {
      const float tot = ((part[0][l] + part[1][l]) + part[2][l]) + part[3][l];
      if (isb) db[m] += ((bpart[0][l] + bpart[1][l]) + bpart[2][l]) + bpart[3][l];
      // distinct (k, m) map to distinct weight elements within one descriptor, but
      // several descriptors (stride phases) may run back to back on the stream.
      float* o = dw + (int64_t)m * w_ms + woff[k];
      if (ADD) *o += tot; else *o = tot;
    }
    __syncthreads();
  }
}


// ---------------------------------------------------------------------------------
// weight-gradient MFMA kernel:  dapack[k][m] += sum_p g[m][p] * In(k, p)
// Both operands are contiguous along the reduction axis p in memory, so both are
// staged through LDS ([row][32 positions], stride 33 -> conflict-free operand reads).
// ---------------------------------------------------------------------------------
template <int TM, int TK>
__global__ __launch_bounds__(256) void gemm_wgrad_kernel(const WgradArgs w) {
  constexpr int RM = 64 * TM, RK = 64 * TK, LD = 33;
  __shared__ float gs[RM * LD];
  __shared__ float xs[RK * LD];
  __shared__ rfx_ktab_entry kts[RK];
  const rfx_gemm_desc& d = w.d;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wk = wave & 1;
  // plain order (k tile fastest).  An XCD-grouped order (all tiles of one position split on one XCD) was
  // measured SLOWER (132 -> 116 TF/s-eq at the TCN shape): the splits are too few / too coarse to balance.
  const int zsplit = blockIdx.z;
  const int m0 = blockIdx.y * RM;
  const int k0 = blockIdx.x * RK;
  const int P = d.OA * d.OB;
  for (int i = tid; i < RK; i += 256) {
    rfx_ktab_entry e;
    if (k0 + i < d.Kpad) e = w.ktab[k0 + i];
    else { e.off = 0; e.da = -(1 << 30); e.db = 0; e.flags = 0; }
    kts[i] = e;
  }
  __syncthreads();

  f32x16 acc[TM][TK];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TK; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int t_begin = zsplit * w.tiles_per_block;
  const int t_end = min(t_begin + w.tiles_per_block, w.total_tiles);
  const int prow = tid >> 5;  // 0..7: row group for loads; lane position = tid & 31
  const int pl = tid & 31;
  float gv[RM / 8], xv[RK / 8];
  // Gathers of one 32-position tile into registers.  Invalid lanes read a device 0 (or 1 for the
  // bias column) instead of masking the loaded value, so nothing consumes the result until the
  // LDS store of the NEXT iteration: the loads stay in flight under this tile's MFMAs.
  auto load_tile = [&](int t) {
    const int n = t / w.tiles_per_sample;
    const int j = (t - n * w.tiles_per_sample) * 32 + pl;
    const bool jvalid = j < P;
    const int jj = jvalid ? j : 0;
    const int a = jj / d.OB, b = jj - a * d.OB;
    const int ia0 = a * d.SA, ib0 = b * d.SB;
    const float* inb = w.in + (int64_t)n * d.in_ns + (int64_t)ia0 * d.in_as + (int64_t)ib0 * d.in_bs;
    const float* gb = w.g + (int64_t)n * d.out_ns + (int64_t)(a * d.out_sa + d.out_a0) * d.out_as +
                      (int64_t)(b * d.out_sb + d.out_b0) * d.out_bs;
#pragma unroll
    for (int i = 0; i < RM / 8; ++i) {
      const int m = m0 + prow + 8 * i;
      const bool ok = jvalid & (m < d.M);
      const float* p = ok ? gb + (int64_t)m * d.out_cs : rfx_zero_f32;
      gv[i] = *p;
    }
#pragma unroll
    for (int i = 0; i < RK / 8; ++i) {
      const rfx_ktab_entry e = kts[prow + 8 * i];
      const bool ones = e.flags & 1;
      const bool ok = jvalid & !ones & ((unsigned)(ia0 + e.da) < (unsigned)d.IA) &
                      ((unsigned)(ib0 + e.db) < (unsigned)d.IB);
      const float* p = ok ? inb + e.off : ((ones & jvalid) ? rfx_one_f32 : rfx_zero_f32);
      xv[i] = *p;
    }
  };
  if (t_begin < t_end) load_tile(t_begin);
  for (int t = t_begin; t < t_end; ++t) {
    __syncthreads();  // previous tile's operand reads are done
#pragma unroll
    for (int i = 0; i < RM / 8; ++i) gs[(prow + 8 * i) * LD + pl] = gv[i];
#pragma unroll
    for (int i = 0; i < RK / 8; ++i) xs[(prow + 8 * i) * LD + pl] = xv[i];
    __syncthreads();
    load_tile(t + 1 < t_end ? t + 1 : t);   // unconditional (branch-free): the last tile is re-read
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float av[TM], bv[TK];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) av[tm] = gs[(wm * 32 * TM + tm * 32 + l31) * LD + 2 * kk + h];
#pragma unroll
      for (int tk = 0; tk < TK; ++tk) bv[tk] = xs[(wk * 32 * TK + tk * 32 + l31) * LD + 2 * kk + h];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tk = 0; tk < TK; ++tk)
          acc[tm][tk] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm], bv[tk], acc[tm][tk], 0, 0, 0);
    }
  }
  // D[i = m][j = k] -> dapack[m][k]  (k = lane axis -> 128-byte coalesced atomics)
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tk = 0; tk < TK; ++tk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 * TM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int k = k0 + wk * 32 * TK + tk * 32 + l31;
        if (m < d.M && k < d.K) w.dapack[(int64_t)zsplit * w.split_stride + (int64_t)m * d.Kpad + k] = acc[tm][tk][r];
      }
}

// Thin weight gradient (M <= 8): one wave per k row, lanes along positions.
template <int MM>
__global__ __launch_bounds__(256) void gemm_thin_wgrad_kernel(const WgradArgs w) {
  const rfx_gemm_desc& d = w.d;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int k = blockIdx.x * 4 + wave;
  if (k >= d.K) return;
  const rfx_ktab_entry e = w.ktab[k];
  const bool ones = e.flags & 1;
  const int P = d.OA * d.OB;
  float acc[MM];
#pragma unroll
  for (int m = 0; m < MM; ++m) acc[m] = 0.f;
  const int64_t total = (int64_t)d.N * P;
  const int64_t chunk = (total + gridDim.y - 1) / gridDim.y;
  const int64_t q0 = (int64_t)blockIdx.y * chunk;
  const int64_t q1 = min(q0 + chunk, total);
  for (int64_t q = q0 + lane; q < q1; q += 64) {
    const int n = (int)(q / P);
    const int j = (int)(q - (int64_t)n * P);
    const int a = j / d.OB, b = j - a * d.OB;
    const int ia0 = a * d.SA, ib0 = b * d.SB;
    float xv;
    if (ones) xv = 1.f;
    else {
      const bool ok = (unsigned)(ia0 + e.da) < (unsigned)d.IA && (unsigned)(ib0 + e.db) < (unsigned)d.IB;
      xv = ok ? w.in[(int64_t)n * d.in_ns + (int64_t)ia0 * d.in_as + (int64_t)ib0 * d.in_bs + e.off] : 0.f;
    }
    const float* gb = w.g + (int64_t)n * d.out_ns + (int64_t)(a * d.out_sa + d.out_a0) * d.out_as +
                      (int64_t)(b * d.out_sb + d.out_b0) * d.out_bs;
#pragma unroll
    for (int m = 0; m < MM; ++m)
      if (m < d.M) acc[m] = fmaf(gb[(int64_t)m * d.out_cs], xv, acc[m]);
  }
#pragma unroll
  for (int m = 0; m < MM; ++m) {
    const float s = rfx_wave_sum(acc[m]);
    if (lane == 0 && m < d.M) w.dapack[(int64_t)blockIdx.y * w.split_stride + (int64_t)m * d.Kpad + k] = s;
  }
}


static int unpack_launch(bool add, const float* dapack, const int32_t* woff, int64_t w_ms, int32_t M, int32_t K,
                         int32_t Kpad, float* dw, int32_t splits, int32_t bias_col, float* db, void* stream) {
  if (!dapack || !woff || !dw || M <= 0 || K < 0 || Kpad < K || splits < 1) return -1;
  const int64_t total = (int64_t)K * M;
  if (total == 0) return 0;
  const int64_t stride = (int64_t)M * Kpad;
  const int grid = (int)((total + 63) / 64 < 8192 ? (total + 63) / 64 : 8192);
  if (add) hipLaunchKernelGGL(unpack_add_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, dapack, woff,
                              w_ms, M, K, Kpad, dw, splits, stride, bias_col, db);
  else hipLaunchKernelGGL(unpack_add_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, dapack, woff,
                          w_ms, M, K, Kpad, dw, splits, stride, bias_col, db);
  RFX_CHECK_LAUNCH();
  return 0;
}
extern "C" int rfx_unpack_add(const float* dapack, const int32_t* woff, int64_t w_ms, int32_t M,
                              int32_t K, int32_t Kpad, float* dw, int32_t splits, void* stream) {
  return unpack_launch(true, dapack, woff, w_ms, M, K, Kpad, dw, splits, -1, nullptr, stream);
}
extern "C" int rfx_unpack_add_bias(const float* dapack, const int32_t* woff, int64_t w_ms, int32_t M, int32_t K, int32_t Kpad,
                                   float* dw, int32_t bias_col, float* db, int32_t splits, void* stream) {
  if (!db || K <= 0 || bias_col < 0 || bias_col >= Kpad) return -1;
  return unpack_launch(true, dapack, woff, w_ms, M, K, Kpad, dw, splits, bias_col, db, stream);
}
extern "C" int rfx_unpack_set(const float* dapack, const int32_t* woff, int64_t w_ms, int32_t M,
                              int32_t K, int32_t Kpad, float* dw, int32_t splits, void* stream) {
  return unpack_launch(false, dapack, woff, w_ms, M, K, Kpad, dw, splits, -1, nullptr, stream);
}
// out[m] (+)= sum over splits of dapack[split][m][col] -- the bias-gradient column when the weight part goes through unpack_set
__global__ __launch_bounds__(256) void unpack_col_kernel(const float* __restrict__ dapack, int M, int Kpad, int col, int splits,
                                                         float* __restrict__ out) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  float s = 0.f;
  for (int q = 0; q < splits; ++q) s += dapack[(int64_t)q * M * Kpad + (int64_t)m * Kpad + col];
  out[m] = s;
}
extern "C" int rfx_unpack_col(const float* dapack, int32_t M, int32_t Kpad, int32_t col, int32_t splits, float* out, void* stream) {
  if (!dapack || !out || M <= 0 || col < 0 || col >= Kpad || splits < 1) return -1;
  hipLaunchKernelGGL(unpack_col_kernel, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, dapack, M, Kpad, col, splits, out);
  RFX_CHECK_LAUNCH();
  return 0;
}

// R (channel tiles per wave) is a pure function of M so that host-side packing
// and the kernel agree on Mpad = ceil(M / 32R) * 32R.

extern "C" int rfx_gemm_wgrad(const rfx_gemm_desc* d, const rfx_ktab_entry* ktab, const float* in,
                              const float* gout, float* dapack, int64_t ws_floats, int32_t* splits_out, int32_t prec, void* stream) {
  if (!desc_ok(d) || !ktab || !in || !gout || !dapack || !splits_out) return -1;
  *splits_out = 1;
  if (d->K == 0) return 0;
  const int64_t slice = (int64_t)d->M * d->Kpad;
  const int max_splits = (int)(ws_floats / slice < 1 ? 0 : (ws_floats / slice > 4096 ? 4096 : ws_floats / slice));
  if (max_splits < 1) return -1;
  if (d->in_bf16) return -1;                          // bf16 storage is implemented for the gradient operand only (wide kernel)
  if (d->out_bf16 && (prec != 2 || d->M <= 8 || ((d->out_b0 | d->out_cs | d->out_as | d->out_ns) & 1))) return -1;
  WgradArgs w;
  w.d = *d; w.ktab = ktab; w.in = in; w.g = gout; w.dapack = dapack; w.split_stride = slice;
  const int P = d->OA * d->OB;
  w.tiles_per_sample = (P + 31) / 32;
  w.total_tiles = d->N * w.tiles_per_sample;
  hipStream_t s = (hipStream_t)stream;
  if (d->M <= 8) {
    const int64_t total = (int64_t)d->N * P;
    int splits = (int)(total / 4096 < 1 ? 1 : (total / 4096 > 64 ? 64 : total / 4096));
    splits = min(splits, max_splits);
    *splits_out = splits;
    dim3 grid((d->K + 3) / 4, splits);
    w.tiles_per_block = 0;
    if (d->M <= 1) hipLaunchKernelGGL(gemm_thin_wgrad_kernel<1>, grid, dim3(256), 0, s, w);
    else if (d->M <= 2) hipLaunchKernelGGL(gemm_thin_wgrad_kernel<2>, grid, dim3(256), 0, s, w);
    else if (d->M <= 4) hipLaunchKernelGGL(gemm_thin_wgrad_kernel<4>, grid, dim3(256), 0, s, w);
    else hipLaunchKernelGGL(gemm_thin_wgrad_kernel<8>, grid, dim3(256), 0, s, w);
    RFX_CHECK_LAUNCH();
    return 0;
  }
  // wide-load kernel (gemm_wgrad.h): both operands contiguous and unit-stride along b, quads never straddle an output row
  const int64_t g_span = ((int64_t)(d->M - 1) * llabs(d->out_cs) + ((int64_t)(d->OA - 1) * d->out_sa + d->out_a0) * llabs(d->out_as) +
                          (d->OB - 1 + d->out_b0) + 1) * (d->out_bf16 ? 2 : 4);
  if (prec != 0 && d->SB == 1 && d->in_bs == 1 && d->out_bs == 1 && d->out_sb == 1 &&
      (d->OA == 1 || d->OB % 4 == 0) && d->in_extent > 0 && d->in_extent <= 0x7fffffffLL && g_span <= 0x7fffffffLL &&
      d->out_cs >= 0 && d->out_as >= 0) {
    w.in_bytes = (uint32_t)d->in_extent; w.g_bytes = (uint32_t)g_span;
    const bool r96 = d->M > 64 && ((d->M + 95) / 96) * 96 < ((d->M + 127) / 128) * 128;
    // 96 x 256 tiles (bf16 mode: the hi + lo images of the split mode would need 101 KB of LDS) when K has >= 2 of them:
    // twice the MFMA work per staged element and half the re-reads of g
    const int shape = r96 ? ((prec == 2 && d->K > 256) ? 5 : 0) : d->M <= 32 ? (d->K > 128 ? 1 : 2) : d->M > 64 ? 3 : 4;
    const int rm = (shape == 0 || shape == 5) ? 96 : shape <= 2 ? 32 : shape == 3 ? 128 : 64;
    const int rk = (shape == 1 || shape == 5) ? 256 : 128;
    const int mt = (d->M + rm - 1) / rm, kt = (d->K + rk - 1) / rk;
    const int chunks = d->N * d->OA * ((d->OB + 63) / 64);          // 64-position chunks of one (n, a) row
    int splits = max(1, 2048 / (mt * kt));
    const int min_chunks = (int64_t)mt * kt * (chunks / 32) >= 512 ? 32 : 8;
    splits = min(min(splits, max_splits), max(1, chunks / min_chunks));
    w.total_tiles = chunks;
    w.tiles_per_block = (chunks + splits - 1) / splits;
    splits = (chunks + w.tiles_per_block - 1) / w.tiles_per_block;
    w.kt = kt; w.mt = mt; w.splits = splits;
    *splits_out = splits;
    // every (k, m) tile of a position split re-reads the same g rows / input rows: group them behind one L2
    // (measured on the Demucs B=64 step, bf16: 44.3 -> 40.2 ms of weight-gradient launches)
    w.xcd_grouped = splits >= 8;
    dim3 grid = w.xcd_grouped ? dim3(((splits + 7) / 8) * 8 * kt * mt, 1, 1) : dim3(kt, mt, splits);
    return prec == 1 ? rfx_launch_wgrad_wide_bf3(w, shape, grid, s) : rfx_launch_wgrad_wide_bf16(w, shape, grid, s);
  }
  // strided plans (the time branch's encoder convs): the 32-position kernel takes a bf16 gradient operand with two-byte loads.
  // num_records of its sample descriptors is 2^31 - 1 bytes: the operand's span past a sample base must stay below that
  if (d->out_bf16 && (prec != 2 || g_span > 0x7fffffffLL)) return -1;
  const bool narrow = prec != 0 && d->M <= 32;            // 32 x (128 tk) tiles, waves 1 x 4 (see gemm_wgrad_bf3_kernel)
  // 96-row tiles (waves 1 x 4, three 32-row MFMA tiles each) when they pad M less than 128-row ones: M = 96, 192, 288
  const bool rows96 = prec != 0 && d->M > 64 && d->K > 64 && ((d->M + 95) / 96) * 96 < ((d->M + 127) / 128) * 128;
  const int tm = d->M > 64 ? 2 : 1;
  const int tk = narrow ? (d->K > 128 ? 2 : 1) : rows96 ? 1 : (d->K > 64 ? 2 : 1);
  const int rm = narrow ? 32 : rows96 ? 96 : 64 * tm, rk = (narrow || rows96) ? 128 * tk : 64 * tk;
  const int mt = (d->M + rm - 1) / rm, kt = (d->K + rk - 1) / rk;
  // aim for ~2048 workgroups; each should still see >= 16 position tiles
  int splits = max(1, 2048 / (mt * kt));
  // >= 64 position tiles per workgroup when there is plenty of work; short sequences (LSTM / attention projections,
  // P ~ 8-38 k positions) would otherwise launch a few dozen workgroups on 256 CUs: go down to 16 tiles there
  const int min_tiles = (int64_t)mt * kt * (w.total_tiles / 64) >= 512 ? 64 : 16;
  splits = min(min(splits, max_splits), max(1, w.total_tiles / min_tiles));
  w.tiles_per_block = (w.total_tiles + splits - 1) / splits;
  splits = (w.total_tiles + w.tiles_per_block - 1) / w.tiles_per_block;
  w.kt = kt; w.mt = mt; w.splits = splits;
  *splits_out = splits;
  dim3 grid(kt, mt, splits);
  w.xcd_grouped = 0;
  if (prec != 0) {
    // (XCD-grouped block order measured slower for this staging: Demucs B=64 408.0 ms off, 411.8 ms on -- not used here)
    const int shape = rows96 ? 0 : (narrow && tk == 2) ? 1 : narrow ? 2 : (tm == 2 && tk == 2) ? 3 : tm == 2 ? 4 : tk == 2 ? 5 : 6;
    return prec == 1 ? rfx_launch_wgrad_bf3(w, shape, grid, s) : rfx_launch_wgrad_bf16(w, shape, grid, s);
  }
  if (tm == 2 && tk == 2) hipLaunchKernelGGL((gemm_wgrad_kernel<2, 2>), grid, dim3(256), 0, s, w);
  else if (tm == 2) hipLaunchKernelGGL((gemm_wgrad_kernel<2, 1>), grid, dim3(256), 0, s, w);
  else if (tk == 2) hipLaunchKernelGGL((gemm_wgrad_kernel<1, 2>), grid, dim3(256), 0, s, w);
  else hipLaunchKernelGGL((gemm_wgrad_kernel<1, 1>), grid, dim3(256), 0, s, w);
  RFX_CHECK_LAUNCH();
  return 0;
}


