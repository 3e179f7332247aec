// Gather-GEMM family: every convolution on the RemFX hot path (forward, input
// gradient, transposed forward, weight gradient) as one table-driven implicit
// GEMM on the fp32 MFMA pipe of gfx950 (v_mfma_f32_32x32x2_f32, exact fp32).
//
// Layout of the forward kernel (wave64, 4 waves / workgroup):
//   * workgroup tile = (32*R) output channels x 128 output positions; wave w owns
//     positions [32w, 32w+32) and all R channel tiles -> the B operand (gathered
//     input samples) goes global -> VGPR directly, one dword per lane per MFMA
//     k-pair, coalesced along the contiguous position axis; no LDS round trip.
//   * the A operand (packed weights [Kpad][Mpad], M contiguous) is staged in LDS
//     with 16-byte loads, double buffered, one barrier per 16-deep K step.
//   * per-k tap metadata (offset + displacement for the bounds test) is wave
//     uniform and read through the scalar cache.
// Replaces F.conv1d/conv2d/conv_transpose1d/2d call sites: tcn.py:50,54,129;
// HDemucs / DCUNet / Cnn14 stacks (models.py:319,358; classifier.py:271-272).
#include "common.h"

struct FwdArgs {
  rfx_gemm_desc d;
  const float* apack;
  const rfx_ktab_entry* ktab;
  const float* in;
  float* out;
  rfx_epilogue e;
  const float* apack2;
  const rfx_ktab_entry* ktab2;
  int32_t Kpad2;
  const float* in2;
};

// ---------------------------------------------------------------------------------
// pack / unpack
// ---------------------------------------------------------------------------------
__global__ void pack_a_kernel(const float* __restrict__ w, const int32_t* __restrict__ woff,
                              int64_t w_ms, int M, int K, int Mpad, int Kpad,
                              float* __restrict__ apack) {
  const int64_t total = (int64_t)Kpad * Mpad;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i / Mpad), m = (int)(i % Mpad);
    float v = 0.f;
    if (k < K && m < M) v = w[(int64_t)m * w_ms + woff[k]];
    apack[i] = v;
  }
}

__global__ void unpack_add_kernel(const float* __restrict__ dapack, const int32_t* __restrict__ woff,
                                  int64_t w_ms, int M, int K, int Mpad, float* __restrict__ dw) {
  const int64_t total = (int64_t)K * M;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i / M), m = (int)(i % M);
    // distinct (k, m) map to distinct weight elements within one descriptor, but
    // several descriptors (stride phases) may run back to back on the stream.
    dw[(int64_t)m * w_ms + woff[k]] += dapack[(int64_t)k * Mpad + m];
  }
}

// ---------------------------------------------------------------------------------
// forward MFMA kernel
// ---------------------------------------------------------------------------------
struct LaneCtx {
  const float* inb;  // in + n*in_ns + position offset
  const float* safe; // always-valid address
  int ia0, ib0;
  bool jvalid;
};

__device__ __forceinline__ void load_b8(const rfx_gemm_desc& d, const rfx_ktab_entry* __restrict__ ktab,
                                        int kbase, int h, const LaneCtx& c, float (&b)[8]) {
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    const rfx_ktab_entry e0 = ktab[kbase + 2 * kk];      // wave-uniform -> scalar loads
    const rfx_ktab_entry e1 = ktab[kbase + 2 * kk + 1];
    const int off = h ? e1.off : e0.off;
    const int da = h ? e1.da : e0.da;
    const int db = h ? e1.db : e0.db;
    const bool ok = c.jvalid && (unsigned)(c.ia0 + da) < (unsigned)d.IA &&
                    (unsigned)(c.ib0 + db) < (unsigned)d.IB;
    const float* p = ok ? (c.inb + off) : c.safe;
    const float v = *p;
    b[kk] = ok ? v : 0.f;
  }
}

template <int R>
__device__ __forceinline__ void stage_a_load(const float* __restrict__ apack, int Mpad, int k0, int m0,
                                             int tid, f32x4 (&r)[2]) {
  constexpr int BM = 32 * R;
  constexpr int NV = 16 * BM / 4;  // float4 per tile
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + i * 256;
    if (idx < NV) {
      const int kk = idx / (BM / 4), c4 = idx % (BM / 4);
      r[i] = *reinterpret_cast<const f32x4*>(apack + (int64_t)(k0 + kk) * Mpad + m0 + 4 * c4);
    }
  }
}
template <int R>
__device__ __forceinline__ void stage_a_store(float* as, int tid, const f32x4 (&r)[2]) {
  constexpr int BM = 32 * R;
  constexpr int NV = 16 * BM / 4;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + i * 256;
    if (idx < NV) {
      const int kk = idx / (BM / 4), c4 = idx % (BM / 4);
      *reinterpret_cast<f32x4*>(as + kk * BM + 4 * c4) = r[i];
    }
  }
}

template <int R>
__device__ __forceinline__ void run_phase(const rfx_gemm_desc& d, const float* __restrict__ apack,
                                          const rfx_ktab_entry* __restrict__ ktab, int Kpad, int m0,
                                          const LaneCtx& c, float* as /* [2][16][BM] */,
                                          f32x16 (&acc)[R]) {
  constexpr int BM = 32 * R;
  const int tid = threadIdx.x;
  const int lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int nk = Kpad / 16;
  if (nk == 0) return;
  float bcur[8], bnext[8];
  f32x4 areg[2];
  stage_a_load<R>(apack, d.Mpad, 0, m0, tid, areg);
  load_b8(d, ktab, 0, h, c, bcur);
  stage_a_store<R>(as, tid, areg);
  __syncthreads();
  int cur = 0;
  for (int ks = 0; ks < nk; ++ks) {
    const bool more = ks + 1 < nk;
    if (more) {
      stage_a_load<R>(apack, d.Mpad, (ks + 1) * 16, m0, tid, areg);
      load_b8(d, ktab, (ks + 1) * 16, h, c, bnext);
    }
    const float* a_lds = as + cur * 16 * BM;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
#pragma unroll
      for (int mt = 0; mt < R; ++mt) {
        const float a = a_lds[(2 * kk + h) * BM + mt * 32 + l31];
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bcur[kk], acc[mt], 0, 0, 0);
      }
    }
    if (more) {
      stage_a_store<R>(as + (cur ^ 1) * 16 * BM, tid, areg);
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) bcur[kk] = bnext[kk];
    }
    __syncthreads();
    cur ^= 1;
  }
}

template <int R>
__global__ __launch_bounds__(256) void gemm_fwd_kernel(const FwdArgs g) {
  constexpr int BM = 32 * R;
  __shared__ __attribute__((aligned(16))) float as[2 * 16 * BM];
  const rfx_gemm_desc& d = g.d;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int P = d.OA * d.OB;
  const int n = blockIdx.z;
  const int m0 = blockIdx.y * BM;
  const int j = blockIdx.x * 128 + wave * 32 + l31;
  LaneCtx c;
  c.jvalid = j < P;
  const int jj = c.jvalid ? j : 0;
  const int a = jj / d.OB, b = jj - a * d.OB;
  c.ia0 = a * d.SA;
  c.ib0 = b * d.SB;
  c.safe = g.in;
  c.inb = g.in + (int64_t)n * d.in_ns + (int64_t)c.ia0 * d.in_as + (int64_t)c.ib0 * d.in_bs;

  f32x16 acc[R];
#pragma unroll
  for (int mt = 0; mt < R; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  run_phase<R>(d, g.apack, g.ktab, d.Kpad, m0, c, as, acc);

  const rfx_epilogue& e = g.e;
  const bool two = g.apack2 != nullptr;
  // bias + activation (between the phases when there are two)
#pragma unroll
  for (int mt = 0; mt < R; ++mt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (m < d.M) {
        float v = acc[mt][r];
        if (e.bias) v += e.bias[m];
        if (e.act != RFX_ACT_NONE && !e.bwd) {
          const float s = (e.act == RFX_ACT_PRELU) ? e.act_param[m] : 0.f;
          v = rfx_act_apply(v, e.act, s);
        }
        acc[mt][r] = v;
      }
    }
  }
  if (two) {
    LaneCtx c2 = c;
    if (g.in2) {
      c2.safe = g.in2;
      c2.inb = g.in2 + (c.inb - g.in);
    }
    run_phase<R>(d, g.apack2, g.ktab2, g.Kpad2, m0, c2, as, acc);
  }

  const int64_t opos = (int64_t)(a * d.out_sa + d.out_a0) * d.out_as + (int64_t)(b * d.out_sb + d.out_b0) * d.out_bs;
  float* outp = g.out + (int64_t)n * d.out_ns + opos;
  const float* resp = nullptr;
  if (e.res)
    resp = e.res + (int64_t)n * e.res_ns + (int64_t)(a * d.out_sa + d.out_a0) * e.res_as +
           (int64_t)(b * d.out_sb + d.out_b0) * e.res_bs;
  if (e.bwd) {
    // out = G * act'(pre);  gparam[m] += sum_j G * min(pre, 0)   (PReLU slope gradient)
#pragma unroll
    for (int mt = 0; mt < R; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        float gs = 0.f;
        if (m < d.M && c.jvalid) {
          const float pre = acc[mt][r];
          const float gin = resp[(int64_t)m * e.res_cs];
          const float s = (e.act == RFX_ACT_PRELU) ? e.act_param[m] : 0.f;
          outp[(int64_t)m * d.out_cs] = gin * rfx_act_grad(pre, e.act, s);
          gs = pre < 0.f ? gin * pre : 0.f;
        }
        if (e.gparam) {   // reduce over the 32 position lanes of this half-wave
#pragma unroll
          for (int o = 16; o > 0; o >>= 1) gs += __shfl_xor(gs, o, 64);
          if (l31 == 0 && m < d.M) atomicAdd(e.gparam + m, gs);
        }
      }
    }
    return;
  }
  if (!c.jvalid) return;
#pragma unroll
  for (int mt = 0; mt < R; ++mt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (m < d.M) {
        float v = acc[mt][r];
        if (resp) v += resp[(int64_t)m * e.res_cs];
        if (e.act2 != RFX_ACT_NONE) v = rfx_act_apply(v, e.act2, 0.f);
        outp[(int64_t)m * d.out_cs] = v;
      }
    }
  }
}

// Thin forward kernel: M <= 8 output rows (TCN output conv 256->1, tcn.py:119,129;
// last HDemucs decoders).  HBM-bound: one thread per position, K loop with
// wave-uniform weights, coalesced gathers.
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
  const float* inb = g.in + (int64_t)n * d.in_ns + (int64_t)ia0 * d.in_as + (int64_t)ib0 * d.in_bs;
  float acc[MM];
#pragma unroll
  for (int m = 0; m < MM; ++m) acc[m] = 0.f;
  const rfx_ktab_entry* __restrict__ kt = g.ktab;
  const float* __restrict__ ap = g.apack;
  for (int k0 = 0; k0 < d.K; k0 += 4) {
    float bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const rfx_ktab_entry e = kt[k0 + u];  // Kpad is a multiple of 16: always readable
      const bool ok = jvalid && (unsigned)(ia0 + e.da) < (unsigned)d.IA &&
                      (unsigned)(ib0 + e.db) < (unsigned)d.IB;
      const float* p = ok ? inb + e.off : g.in;
      const float v = *p;
      bv[u] = ok ? v : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int m = 0; m < MM; ++m) acc[m] = fmaf(ap[(int64_t)(k0 + u) * d.Mpad + m], bv[u], acc[m]);
  }
  if (!jvalid) return;
  const rfx_epilogue& e = g.e;
  const int64_t opos = (int64_t)(a * d.out_sa + d.out_a0) * d.out_as + (int64_t)(b * d.out_sb + d.out_b0) * d.out_bs;
  float* outp = g.out + (int64_t)n * d.out_ns + opos;
#pragma unroll
  for (int m = 0; m < MM; ++m) {
    if (m < d.M) {
      float v = acc[m];
      if (e.bias) v += e.bias[m];
      if (e.act != RFX_ACT_NONE) v = rfx_act_apply(v, e.act, e.act == RFX_ACT_PRELU ? e.act_param[m] : 0.f);
      if (e.res)
        v += e.res[(int64_t)n * e.res_ns + (int64_t)m * e.res_cs +
                   (int64_t)(a * d.out_sa + d.out_a0) * e.res_as + (int64_t)(b * d.out_sb + d.out_b0) * e.res_bs];
      if (e.act2 != RFX_ACT_NONE) v = rfx_act_apply(v, e.act2, 0.f);
      outp[(int64_t)m * d.out_cs] = v;
    }
  }
}

// ---------------------------------------------------------------------------------
// weight-gradient MFMA kernel:  dapack[k][m] += sum_p g[m][p] * In(k, p)
// Both operands are contiguous along the reduction axis p in memory, so both are
// staged through LDS ([row][32 positions], stride 33 -> conflict-free operand reads).
// ---------------------------------------------------------------------------------
struct WgradArgs {
  rfx_gemm_desc d;
  const rfx_ktab_entry* ktab;
  const float* in;
  const float* g;
  float* dapack;
  int tiles_per_sample;  // ceil(P / 32)
  int total_tiles;       // N * tiles_per_sample
  int tiles_per_block;
};

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

  const int t_begin = blockIdx.z * w.tiles_per_block;
  const int t_end = min(t_begin + w.tiles_per_block, w.total_tiles);
  const int prow = tid >> 5;  // 0..7: row group for loads; lane position = tid & 31
  const int pl = tid & 31;
  for (int t = t_begin; t < t_end; ++t) {
    const int n = t / w.tiles_per_sample;
    const int j = (t - n * w.tiles_per_sample) * 32 + pl;
    const bool jvalid = j < P;
    const int jj = jvalid ? j : 0;
    const int a = jj / d.OB, b = jj - a * d.OB;
    const int ia0 = a * d.SA, ib0 = b * d.SB;
    const float* inb = w.in + (int64_t)n * d.in_ns + (int64_t)ia0 * d.in_as + (int64_t)ib0 * d.in_bs;
    const float* gb = w.g + (int64_t)n * d.out_ns + (int64_t)(a * d.out_sa + d.out_a0) * d.out_as +
                      (int64_t)(b * d.out_sb + d.out_b0) * d.out_bs;
    float gv[RM / 8], xv[RK / 8];
#pragma unroll
    for (int i = 0; i < RM / 8; ++i) {
      const int m = m0 + prow + 8 * i;
      const bool ok = jvalid && m < d.M;
      const float* p = ok ? gb + (int64_t)m * d.out_cs : w.g;
      const float v = *p;
      gv[i] = ok ? v : 0.f;
    }
#pragma unroll
    for (int i = 0; i < RK / 8; ++i) {
      const rfx_ktab_entry e = kts[prow + 8 * i];
      const bool ones = e.flags & 1;
      const bool ok = jvalid && !ones && (unsigned)(ia0 + e.da) < (unsigned)d.IA &&
                      (unsigned)(ib0 + e.db) < (unsigned)d.IB;
      const float* p = ok ? inb + e.off : w.in;
      const float v = *p;
      xv[i] = ok ? v : ((ones && jvalid) ? 1.f : 0.f);
    }
    __syncthreads();  // previous tile's operand reads are done
#pragma unroll
    for (int i = 0; i < RM / 8; ++i) gs[(prow + 8 * i) * LD + pl] = gv[i];
#pragma unroll
    for (int i = 0; i < RK / 8; ++i) xs[(prow + 8 * i) * LD + pl] = xv[i];
    __syncthreads();
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
  // D[i = m][j = k] -> dapack[k][m]
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tk = 0; tk < TK; ++tk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 * TM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int k = k0 + wk * 32 * TK + tk * 32 + l31;
        if (m < d.M && k < d.K) atomicAdd(w.dapack + (int64_t)k * d.Mpad + m, acc[tm][tk][r]);
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
    if (lane == 0 && m < d.M) atomicAdd(w.dapack + (int64_t)k * d.Mpad + m, s);
  }
}

// ---------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------
static bool desc_ok(const rfx_gemm_desc* d) {
  return d && d->N > 0 && d->M > 0 && d->K >= 0 && d->OA > 0 && d->OB > 0 && d->Mpad % 4 == 0 &&
         d->Kpad % 16 == 0 && d->Kpad >= d->K && d->Mpad >= d->M;
}

extern "C" int rfx_abi_version(void) { return RFX_ABI_VERSION; }

extern "C" int rfx_pack_a(const float* w, const int32_t* woff, int64_t w_ms, int32_t M, int32_t K,
                          int32_t Mpad, int32_t Kpad, float* apack, void* stream) {
  if (!w || !woff || !apack || M <= 0 || K < 0 || Mpad < M || Kpad < K) return -1;
  const int64_t total = (int64_t)Kpad * Mpad;
  if (total == 0) return 0;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_a_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, woff, w_ms, M, K,
                     Mpad, Kpad, apack);
  RFX_CHECK_LAUNCH();
  return 0;
}

extern "C" int rfx_unpack_add(const float* dapack, const int32_t* woff, int64_t w_ms, int32_t M,
                              int32_t K, int32_t Mpad, float* dw, void* stream) {
  if (!dapack || !woff || !dw || M <= 0 || K < 0 || Mpad < M) return -1;
  const int64_t total = (int64_t)K * M;
  if (total == 0) return 0;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(unpack_add_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dapack, woff,
                     w_ms, M, K, Mpad, dw);
  RFX_CHECK_LAUNCH();
  return 0;
}

// R (channel tiles per wave) is a pure function of M so that host-side packing
// and the kernel agree on Mpad = ceil(M / 32R) * 32R.
static int pick_r(int M) {
  if (M <= 8) return 0;  // thin path
  int best = 4, best_pad = ((M + 127) / 128) * 128;
  for (int r = 3; r >= 2; --r) {
    const int bm = 32 * r, pad = ((M + bm - 1) / bm) * bm;
    if (pad < best_pad) { best = r; best_pad = pad; }
  }
  if (M <= 32) best = 1;
  return best;
}

extern "C" int rfx_gemm_pick_r(int32_t M) { return pick_r(M); }

extern "C" int rfx_gemm_fwd(const rfx_gemm_desc* d, const float* apack, const rfx_ktab_entry* ktab,
                            const float* in, float* out, const rfx_epilogue* epi, const float* apack2,
                            const rfx_ktab_entry* ktab2, int32_t K2, int32_t Kpad2, const float* in2,
                            void* stream) {
  if (!desc_ok(d) || !apack || !ktab || !in || !out) return -1;
  if ((apack2 != nullptr) != (ktab2 != nullptr)) return -1;
  if (apack2 && (Kpad2 % 16 != 0 || Kpad2 < K2)) return -1;
  FwdArgs g;
  g.d = *d;
  g.apack = apack; g.ktab = ktab; g.in = in; g.out = out;
  if (epi) g.e = *epi;
  else { g.e = rfx_epilogue{}; }
  g.apack2 = apack2; g.ktab2 = ktab2; g.Kpad2 = apack2 ? Kpad2 : 0; g.in2 = in2;
  const int P = d->OA * d->OB;
  const int r = pick_r(d->M);
  hipStream_t s = (hipStream_t)stream;
  if (r == 0) {
    if (apack2 || g.e.bwd) return -1;
    dim3 grid((P + 255) / 256, d->N);
    if (d->M <= 1) hipLaunchKernelGGL(gemm_thin_fwd_kernel<1>, grid, dim3(256), 0, s, g);
    else if (d->M <= 2) hipLaunchKernelGGL(gemm_thin_fwd_kernel<2>, grid, dim3(256), 0, s, g);
    else if (d->M <= 4) hipLaunchKernelGGL(gemm_thin_fwd_kernel<4>, grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL(gemm_thin_fwd_kernel<8>, grid, dim3(256), 0, s, g);
    RFX_CHECK_LAUNCH();
    return 0;
  }
  const int bm = 32 * r;
  if (d->Mpad % bm != 0) return -1;
  dim3 grid((P + 127) / 128, d->Mpad / bm, d->N);
  switch (r) {
    case 1: hipLaunchKernelGGL(gemm_fwd_kernel<1>, grid, dim3(256), 0, s, g); break;
    case 2: hipLaunchKernelGGL(gemm_fwd_kernel<2>, grid, dim3(256), 0, s, g); break;
    case 3: hipLaunchKernelGGL(gemm_fwd_kernel<3>, grid, dim3(256), 0, s, g); break;
    default: hipLaunchKernelGGL(gemm_fwd_kernel<4>, grid, dim3(256), 0, s, g); break;
  }
  RFX_CHECK_LAUNCH();
  return 0;
}

extern "C" int rfx_gemm_wgrad(const rfx_gemm_desc* d, const rfx_ktab_entry* ktab, const float* in,
                              const float* gout, float* dapack, void* stream) {
  if (!desc_ok(d) || !ktab || !in || !gout || !dapack) return -1;
  if (d->K == 0) return 0;
  WgradArgs w;
  w.d = *d; w.ktab = ktab; w.in = in; w.g = gout; w.dapack = dapack;
  const int P = d->OA * d->OB;
  w.tiles_per_sample = (P + 31) / 32;
  w.total_tiles = d->N * w.tiles_per_sample;
  hipStream_t s = (hipStream_t)stream;
  if (d->M <= 8) {
    const int64_t total = (int64_t)d->N * P;
    int splits = (int)(total / 4096 < 1 ? 1 : (total / 4096 > 64 ? 64 : total / 4096));
    dim3 grid((d->K + 3) / 4, splits);
    w.tiles_per_block = 0;
    if (d->M <= 1) hipLaunchKernelGGL(gemm_thin_wgrad_kernel<1>, grid, dim3(256), 0, s, w);
    else if (d->M <= 2) hipLaunchKernelGGL(gemm_thin_wgrad_kernel<2>, grid, dim3(256), 0, s, w);
    else if (d->M <= 4) hipLaunchKernelGGL(gemm_thin_wgrad_kernel<4>, grid, dim3(256), 0, s, w);
    else hipLaunchKernelGGL(gemm_thin_wgrad_kernel<8>, grid, dim3(256), 0, s, w);
    RFX_CHECK_LAUNCH();
    return 0;
  }
  const int tm = d->M > 64 ? 2 : 1;
  const int tk = d->K > 64 ? 2 : 1;
  const int mt = (d->M + 64 * tm - 1) / (64 * tm), kt = (d->K + 64 * tk - 1) / (64 * tk);
  // aim for ~2048 workgroups; each should still see >= 16 position tiles
  int splits = max(1, 2048 / (mt * kt));
  splits = min(splits, max(1, w.total_tiles / 16));
  w.tiles_per_block = (w.total_tiles + splits - 1) / splits;
  splits = (w.total_tiles + w.tiles_per_block - 1) / w.tiles_per_block;
  dim3 grid(kt, mt, splits);
  if (tm == 2 && tk == 2) hipLaunchKernelGGL((gemm_wgrad_kernel<2, 2>), grid, dim3(256), 0, s, w);
  else if (tm == 2) hipLaunchKernelGGL((gemm_wgrad_kernel<2, 1>), grid, dim3(256), 0, s, w);
  else if (tk == 2) hipLaunchKernelGGL((gemm_wgrad_kernel<1, 2>), grid, dim3(256), 0, s, w);
  else hipLaunchKernelGGL((gemm_wgrad_kernel<1, 1>), grid, dim3(256), 0, s, w);
  RFX_CHECK_LAUNCH();
  return 0;
}
