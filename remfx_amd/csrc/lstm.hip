// LSTM recurrence (forward and backward through time) as ONE persistent launch per layer:
// the BiLSTM of HDemucs' DConv (torchaudio _BLSTM via models.py:319) and Open-Unmix's 3-layer
// BiLSTM (models.py:297-298).  Replaces cuDNN/MIOpen-style "one GEMM + one pointwise kernel per
// time step" (2624 GEMM launches per Demucs training step in the rocprof r01 traces).
//
// Everything around the recurrence is a plain gather-GEMM on channel-major tensors (1, C, T*Bn):
// input projections xp = W_ih x + b, and after the backward sweep dX, dW_ih, dW_hh, db.
// The sweep itself:
//   * grid = (batch tiles of 32 sequences, 2 directions); a workgroup has H/32 waves and owns ALL
//     hidden units of its sequences, so time steps need one workgroup barrier, never a grid sync;
//   * gates^T[4H x 32] = xp[t] + W_hh[4H x H] . h^T[H x 32] on v_mfma_f32_32x32x16_bf16 with the
//     bf16x3 split (hi.hi + hi.lo + lo.hi, fp32 accumulate); MFMA rows = gate units, columns =
//     sequences, so every global access is coalesced along the sequence axis of the channel-major
//     tensors and a lane owns (16 units x 1 sequence) of all four gates -> the cell update is local;
//   * W_hh streams from L2 every step as pre-packed MFMA A fragments (pack kernel below);
//     h_{t-1} (resp. the gate gradients) is the B operand, kept in LDS as bf16 hi/lo rows.
#include "common.h"

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

// fwdA[ub][g][ks][lane]: W_hh[g*H + 32ub + (lane&31)][16ks + 8(lane>>5) + q], q = 0..7   (hi array, then lo array)
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
        const int ks = (int)(r % nks), g = (int)((r / nks) % 4), ub = (int)(r / (4 * nks));
        v = whh[(int64_t)(g * H + 32 * ub + (lane & 31)) * H + 16 * ks + 8 * (lane >> 5) + q];
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

struct LstmArgs {
  const float* xp;      // [2][4H][P]   input projections (+ both biases), P = T*Bn, position = t*Bn + b
  const uint4* packA;   // per direction: fwdA (hi, lo) then bwdA (hi, lo); dir stride in uint4 = pack_stride
  float* out;           // [2H][P]      channel-major output sequence (dir d at rows d*H..)
  float* gates;         // [2][4H][P]   post-activation i, f, g, o (NULL in inference)
  float* cstate;        // [2][H][P]
  const float* gout;    // bwd: [2H][P]
  float* dG;            // bwd: [2][4H][P] gate pre-activation gradients
  int64_t pack_stride;
  int T, Bn, H, P;
};

extern __shared__ __attribute__((aligned(16))) unsigned char lstm_smem[];

struct LstmAFrag { uint4 h[4], l[4]; };

// XPV: the step's input projections are fetched into their own registers at the top of the step and added in
// the cell update, so their HBM latency hides behind the MFMA chain (needs 64 more VGPRs; H <= 256 variants).
// Otherwise they are loaded straight into the accumulators before the previous step's barrier.
template <int MAXT, bool XPV>
__global__ __launch_bounds__(MAXT) void lstm_fwd_kernel(const LstmArgs a) {
  const int H = a.H, Bn = a.Bn, T = a.T, P = a.P;
  const int dir = blockIdx.y, row0 = blockIdx.x * 32;
  const int tid = threadIdx.x, ub = tid >> 6, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int nks = H / 16, LDH = H + 8;
  unsigned short* hbuf = reinterpret_cast<unsigned short*>(lstm_smem);   // [2 bufs][2 hi/lo][32][LDH]
  const int row = row0 + l31;
  const bool rvalid = row < Bn;
  // invalid sequences read (finite) data of sequence 0 and never store: keeps every load unconditional
  const int rowc = rvalid ? row : 0;
  const float* __restrict__ xp = a.xp + (int64_t)dir * 4 * H * P;
  const int nf = (H / 32) * 4 * nks * 64;
  const uint4* __restrict__ Ahi = a.packA + (int64_t)dir * a.pack_stride + ub * 4 * nks * 64 + lane;
  const uint4* __restrict__ Alo = Ahi + nf;
  float* outp = a.out + (int64_t)dir * H * P;
  float* gsave = a.gates ? a.gates + (int64_t)dir * 4 * H * P : nullptr;
  float* csave = a.gates ? a.cstate + (int64_t)dir * H * P : nullptr;
  const uint32_t uP = (uint32_t)P, HP = (uint32_t)H * uP;
  const uint32_t ubase = (uint32_t)(32 * ub + 4 * hh) * uP + (uint32_t)rowc;   // + ((r&3) + 8(r>>2)) * P + t*Bn
  auto loadA = [&](LstmAFrag& f, int ks) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f.h[g] = Ahi[(g * nks + ks) * 64];
      f.l[g] = Alo[(g * nks + ks) * 64];
    }
  };
  auto mma = [&](f32x16* acc, const LstmAFrag& f, const unsigned short* hb, int ks) {
    const int off = l31 * LDH + 16 * ks + 8 * hh;
    const bf16x8 bh = *reinterpret_cast<const bf16x8*>(hb + off);
    const bf16x8 bl = *reinterpret_cast<const bf16x8*>(hb + 32 * LDH + off);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const bf16x8 ah = __builtin_bit_cast(bf16x8, f.h[g]);
      const bf16x8 al = __builtin_bit_cast(bf16x8, f.l[g]);
      acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[g], 0, 0, 0);
      acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[g], 0, 0, 0);
      acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[g], 0, 0, 0);
    }
  };
  float c[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  f32x16 acc[4];
  LstmAFrag f0, f1;
  loadA(f0, 0);
  if (!XPV) {
    const uint32_t o0 = ubase + (uint32_t)((dir == 0 ? 0 : T - 1) * Bn);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[g][r] = xp[o0 + (uint32_t)g * HP + (uint32_t)((r & 3) + 8 * (r >> 2)) * uP];
  }
  for (int s = 0; s < T; ++s) {
    const int t = dir == 0 ? s : T - 1 - s;
    const uint32_t o0 = ubase + (uint32_t)(t * Bn);
    float xpv[XPV ? 4 : 1][16];
    if (XPV) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          xpv[g][r] = xp[o0 + (uint32_t)g * HP + (uint32_t)((r & 3) + 8 * (r >> 2)) * uP];
          acc[g][r] = 0.f;
        }
    }
    if (s > 0) {
      const unsigned short* hb = hbuf + ((s - 1) & 1) * 2 * 32 * LDH;
      for (int ks = 0; ks < nks; ks += 2) {          // nks is even; weights stay two k-steps ahead
        loadA(f1, ks + 1);
        mma(acc, f0, hb, ks);
        loadA(f0, ks + 2 == nks ? 0 : ks + 2);       // wraps to the next step's first fragment
        mma(acc, f1, hb, ks + 1);
      }
    }
    unsigned short* hw = hbuf + (s & 1) * 2 * 32 * LDH + l31 * LDH + 32 * ub + 4 * hh;
    // next step's time index (clamped on the last step: a harmless re-read)
    const int tn = s + 1 < T ? (dir == 0 ? s + 1 : T - 2 - s) : t;
    const uint32_t on = ubase + (uint32_t)(tn * Bn);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int du = (r & 3) + 8 * (r >> 2);
      float p0 = acc[0][r], p1 = acc[1][r], p2 = acc[2][r], p3 = acc[3][r];
      if (XPV) { p0 += xpv[0][r]; p1 += xpv[1][r]; p2 += xpv[2][r]; p3 += xpv[3][r]; }
      const float ig = lstm_sigmoid(p0), fg = lstm_sigmoid(p1);
      const float gg = lstm_tanh(p2), og = lstm_sigmoid(p3);
      c[r] = fg * c[r] + ig * gg;
      const float hv = og * lstm_tanh(c[r]);
      unsigned short vh, vl;
      split_hi_lo(hv, vh, vl);
      hw[du] = vh;
      hw[32 * LDH + du] = vl;
      if (rvalid) {
        const uint32_t o = o0 + (uint32_t)du * uP;
        outp[o] = hv;
        if (gsave) {
          gsave[o] = ig; gsave[o + HP] = fg; gsave[o + 2 * HP] = gg; gsave[o + 3 * HP] = og;
          csave[o] = c[r];
        }
      }
      if (!XPV) {
        const uint32_t o = on + (uint32_t)du * uP;
        acc[0][r] = xp[o]; acc[1][r] = xp[o + HP]; acc[2][r] = xp[o + 2 * HP]; acc[3][r] = xp[o + 3 * HP];
      }
    }
    __syncthreads();
  }
}

// backward through time.  RB = sequences per workgroup (32, or 16 when 4H bf16 hi/lo rows of 32 would not fit LDS)
template <int RB, int MAXT>
__global__ __launch_bounds__(MAXT) void lstm_bwd_kernel(const LstmArgs a) {
  const int H = a.H, Bn = a.Bn, T = a.T, P = a.P;
  const int dir = blockIdx.y, row0 = blockIdx.x * RB;
  const int tid = threadIdx.x, ub = tid >> 6, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int nks4 = 4 * H / 16, LDG = 4 * H + 8;
  unsigned short* gbuf = reinterpret_cast<unsigned short*>(lstm_smem);   // [2 hi/lo][RB][LDG]
  const int lr = l31 & (RB - 1);
  const int row = row0 + lr;
  const bool rvalid = (l31 < RB) && row < Bn;
  const int nks = H / 16;
  const int nf = (H / 32) * 4 * nks * 64, nb = (H / 32) * nks4 * 64;
  const uint4* __restrict__ Ahi = a.packA + (int64_t)dir * a.pack_stride + 2 * nf + ub * nks4 * 64;
  const uint4* __restrict__ Alo = Ahi + nb;
  const float* __restrict__ gates = a.gates + (int64_t)dir * 4 * H * P;
  const float* __restrict__ cst = a.cstate + (int64_t)dir * H * P;
  float* __restrict__ dG = a.dG + (int64_t)dir * 4 * H * P;
  const float* __restrict__ goutp = a.gout + (int64_t)dir * H * P;
  const uint32_t uP = (uint32_t)P, HP = (uint32_t)H * uP;
  const uint32_t ubase = (uint32_t)(32 * ub + 4 * hh) * uP + (uint32_t)row;
  unsigned short* gr = gbuf + lr * LDG + 32 * ub + 4 * hh;
  float dcc[16];
  f32x16 dhr;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dcc[r] = 0.f; dhr[r] = 0.f; }
  for (int s = T - 1; s >= 0; --s) {            // reverse of the forward processing order
    const int t = dir == 0 ? s : T - 1 - s;
    const int tprev = dir == 0 ? t - 1 : t + 1;   // time index of the forward sweep's previous step
    const uint32_t o0 = ubase + (uint32_t)(t * Bn);
    const uint32_t op0 = ubase + (uint32_t)((s > 0 ? tprev : t) * Bn);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int du = (r & 3) + 8 * (r >> 2);
      float di = 0.f, df = 0.f, dg = 0.f, dO = 0.f;
      if (rvalid) {
        const uint32_t o = o0 + (uint32_t)du * uP;
        const float ig = gates[o], fg = gates[o + HP], gg = gates[o + 2 * HP], og = gates[o + 3 * HP];
        const float ct = cst[o];
        float cp = cst[op0 + (uint32_t)du * uP];
        cp = s > 0 ? cp : 0.f;
        const float dh = goutp[o] + dhr[r];
        const float th = lstm_tanh(ct);
        dO = dh * th * og * (1.f - og);
        const float dc = dh * og * (1.f - th * th) + dcc[r];
        di = dc * gg * ig * (1.f - ig);
        df = dc * cp * fg * (1.f - fg);
        dg = dc * ig * (1.f - gg * gg);
        dcc[r] = dc * fg;
        dG[o] = di; dG[o + HP] = df; dG[o + 2 * HP] = dg; dG[o + 3 * HP] = dO;
      }
      if (l31 < RB) {
        unsigned short vh, vl;
        split_hi_lo(di, vh, vl); gr[du] = vh; gr[RB * LDG + du] = vl;
        split_hi_lo(df, vh, vl); gr[H + du] = vh; gr[RB * LDG + H + du] = vl;
        split_hi_lo(dg, vh, vl); gr[2 * H + du] = vh; gr[RB * LDG + 2 * H + du] = vl;
        split_hi_lo(dO, vh, vl); gr[3 * H + du] = vh; gr[RB * LDG + 3 * H + du] = vl;
      }
    }
    if (s == 0) break;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) dhr[r] = 0.f;
    for (int ks = 0; ks < nks4; ++ks) {
      const int off = lr * LDG + 16 * ks + 8 * hh;
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(gbuf + off);
      const bf16x8 bl = *reinterpret_cast<const bf16x8*>(gbuf + RB * LDG + off);
      const bf16x8 ah = __builtin_bit_cast(bf16x8, Ahi[ks * 64 + lane]);
      const bf16x8 al = __builtin_bit_cast(bf16x8, Alo[ks * 64 + lane]);
      dhr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, dhr, 0, 0, 0);
      dhr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, dhr, 0, 0, 0);
      dhr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, dhr, 0, 0, 0);
    }
    __syncthreads();
  }
}

static int64_t lstm_pack_uint4_per_dir(int H) {
  const int64_t nf = (int64_t)(H / 32) * 4 * (H / 16) * 64, nb = (int64_t)(H / 32) * (4 * H / 16) * 64;
  return 2 * nf + 2 * nb;
}

extern "C" int rfx_lstm_pack_bytes(int32_t H) {
  if (H <= 0 || H % 32 || H > 512) return -1;
  return (int)(lstm_pack_uint4_per_dir(H) * 16);
}

extern "C" int rfx_lstm_pack(const float* whh, int32_t H, void* pack, void* stream) {
  if (!whh || !pack || H <= 0 || H % 32 || H > 512) return -1;
  const int64_t nf = (int64_t)(H / 32) * 4 * (H / 16) * 64;
  uint4* p = reinterpret_cast<uint4*>(pack);
  const int64_t total = lstm_pack_uint4_per_dir(H) / 2;
  const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(lstm_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, whh, H, p, p + 2 * nf);
  RFX_CHECK_LAUNCH();
  return 0;
}

extern "C" int rfx_lstm_fwd(const float* xp, const void* pack, int32_t T, int32_t Bn, int32_t H, float* out,
                            float* gates, float* cstate, void* stream) {
  if (!xp || !pack || !out || T <= 0 || Bn <= 0 || H <= 0 || H % 32 || H > 512) return -1;
  if ((gates == nullptr) != (cstate == nullptr)) return -1;
  if ((int64_t)4 * H * T * Bn >= ((int64_t)1 << 31)) return -1;
  LstmArgs a{};
  a.xp = xp; a.packA = reinterpret_cast<const uint4*>(pack); a.out = out; a.gates = gates; a.cstate = cstate;
  a.pack_stride = lstm_pack_uint4_per_dir(H); a.T = T; a.Bn = Bn; a.H = H; a.P = T * Bn;
  const size_t smem = (size_t)2 * 2 * 32 * (H + 8) * sizeof(unsigned short);
  const dim3 grid((Bn + 31) / 32, 2), block(64 * (H / 32));
  hipStream_t s = (hipStream_t)stream;
#define RFX_LSTM_FWD(MAXT, XPV)                                                                                  \
  do {                                                                                                           \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_fwd_kernel<MAXT, XPV>),                           \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return -3;     \
    hipLaunchKernelGGL((lstm_fwd_kernel<MAXT, XPV>), grid, block, smem, s, a);                                   \
  } while (0)
  if (block.x <= 512) RFX_LSTM_FWD(512, true);
  else if (block.x <= 768) RFX_LSTM_FWD(768, false);
  else RFX_LSTM_FWD(1024, false);
#undef RFX_LSTM_FWD
  RFX_CHECK_LAUNCH();
  return 0;
}

extern "C" int rfx_lstm_bwd(const float* gout, const void* pack, const float* gates, const float* cstate, int32_t T,
                            int32_t Bn, int32_t H, float* dG, void* stream) {
  if (!gout || !pack || !gates || !cstate || !dG || T <= 0 || Bn <= 0 || H <= 0 || H % 32 || H > 512) return -1;
  if ((int64_t)4 * H * T * Bn >= ((int64_t)1 << 31)) return -1;
  LstmArgs a{};
  a.gout = gout; a.packA = reinterpret_cast<const uint4*>(pack); a.gates = const_cast<float*>(gates);
  a.cstate = const_cast<float*>(cstate); a.dG = dG;
  a.pack_stride = lstm_pack_uint4_per_dir(H); a.T = T; a.Bn = Bn; a.H = H; a.P = T * Bn;
  const int rb = (size_t)2 * 32 * (4 * H + 8) * 2 <= 150 * 1024 ? 32 : 16;
  const size_t smem = (size_t)2 * rb * (4 * H + 8) * sizeof(unsigned short);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((Bn + rb - 1) / rb, 2), block(64 * (H / 32));
#define RFX_LSTM_BWD(RB, MAXT)                                                                                   \
  do {                                                                                                           \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_bwd_kernel<RB, MAXT>),                            \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return -3;     \
    hipLaunchKernelGGL((lstm_bwd_kernel<RB, MAXT>), grid, block, smem, s, a);                                    \
  } while (0)
  if (rb == 32) {
    if (block.x <= 512) RFX_LSTM_BWD(32, 512);
    else RFX_LSTM_BWD(32, 1024);
  } else {
    if (block.x <= 768) RFX_LSTM_BWD(16, 768);
    else RFX_LSTM_BWD(16, 1024);
  }
#undef RFX_LSTM_BWD
  RFX_CHECK_LAUNCH();
  return 0;
}
