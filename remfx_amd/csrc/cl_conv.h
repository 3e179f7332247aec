// Kernel template of rfx_cl_conv (csrc/cl_conv.hip holds the description and the entry point); instantiated per epilogue mode in
// csrc/cl_conv_m_*.hip: the modes compile in parallel, and a kernel carries one epilogue (no run-time dispatch, no dead operands).
#pragma once
#include "cl_common.h"

// ablation builds (scripts/build_abl.py, dev only): 1 no DMA, 2 no fragment reads, 4 no MFMA, 8 no epilogue, 16 no unit barrier
#ifndef RFX_CLC_DBG_BUILD
#define RFX_CLC_DBG_BUILD 0
#endif

struct ClConvK {
  rfx_cl_conv_desc d;
  int32_t ptiles, chunk, MG, tpr;
  uint32_t in_bytes;      // byte extent of one sample of `in`
};

template <int RW, int NT, int WM, int NTC, int KS, int DA, int DB, bool HALO>
struct ClConvCfg {
  static constexpr int WN = 8 / WM, BM = 32 * RW * WM, MT = RW * WM;
  static constexpr int A_KB = NTC * KS * MT;                           // 1-KiB A fragments per unit
  static constexpr int PA = (A_KB + 3) / 4;                            // A pieces per loader wave (waves 4..7) and unit
  static constexpr int SLOTS = HALO ? 272 : 256, CORE0 = HALO ? 16 : 0, PLANE = SLOTS * 32;
  static constexpr int PB = KS * (HALO ? 3 : 2);                       // B pieces per loader wave (waves 0..3) and unit
  static constexpr int A_UNIT = A_KB * 1024, B_UNIT = KS * PLANE;
  static constexpr int RING = DA * A_UNIT + DB * B_UNIT;
  static constexpr int EPI_RS = 64 * RW + 8, EPI_RSY = 32 * RW + 8;    // per-wave transpose tiles: row strides in bytes
  static constexpr int EPI_WAVE = 32 * EPI_RS + 32 * EPI_RSY;
  static constexpr int BIAS_OFF = (RING > 8 * EPI_WAVE ? RING : 8 * EPI_WAVE);
  static constexpr int LDS = BIAS_OFF + BM * 4;
  static_assert(NT * WN == 8, "a workgroup covers 256 positions");
  static_assert(DA >= 2 && DB >= 2 && (DA - 2) * PA <= 15 && (DB - 2) * PB <= 15, "counted waits are generated up to vmcnt(15)");
};

// s_waitcnt vmcnt(n) for a wave-uniform run-time n in [0, 15] (the instruction takes a literal); only the last units of a tile
// come here, the steady state waits with a constant
__device__ __forceinline__ void cl_wait_vm(int n) {
  switch (n) {
    case 0: CL_VMCNT(0); break;   case 1: CL_VMCNT(1); break;   case 2: CL_VMCNT(2); break;   case 3: CL_VMCNT(3); break;
    case 4: CL_VMCNT(4); break;   case 5: CL_VMCNT(5); break;   case 6: CL_VMCNT(6); break;   case 7: CL_VMCNT(7); break;
    case 8: CL_VMCNT(8); break;   case 9: CL_VMCNT(9); break;   case 10: CL_VMCNT(10); break; case 11: CL_VMCNT(11); break;
    case 12: CL_VMCNT(12); break; case 13: CL_VMCNT(13); break; case 14: CL_VMCNT(14); break; default: CL_VMCNT(15); break;
  }
}
template <int N>
__device__ __forceinline__ void cl_wait_vm_c() {
  static_assert(N >= 0 && N <= 15, "");
  if constexpr (N == 0) CL_VMCNT(0); else if constexpr (N == 1) CL_VMCNT(1); else if constexpr (N == 2) CL_VMCNT(2);
  else if constexpr (N == 3) CL_VMCNT(3); else if constexpr (N == 4) CL_VMCNT(4); else if constexpr (N == 5) CL_VMCNT(5);
  else if constexpr (N == 6) CL_VMCNT(6); else if constexpr (N == 7) CL_VMCNT(7); else if constexpr (N == 8) CL_VMCNT(8);
  else if constexpr (N == 9) CL_VMCNT(9); else if constexpr (N == 10) CL_VMCNT(10); else if constexpr (N == 11) CL_VMCNT(11);
  else if constexpr (N == 12) CL_VMCNT(12); else if constexpr (N == 13) CL_VMCNT(13); else if constexpr (N == 14) CL_VMCNT(14);
  else CL_VMCNT(15);
}

// slot of relative position p (p in [-8, 264)) in a plane with halo: [left halo 8][right halo 8][core 256]
__device__ __forceinline__ int cl_slot_halo(int p) { return p < 0 ? p + 8 : (p >= 256 ? p - 248 : p + 16); }

typedef uint32_t cl_u32x4 __attribute__((ext_vector_type(4)));
// 16 bytes of a channels-last operand through a buffer descriptor (32-bit byte offset)
__device__ __forceinline__ uint4 cl_bld(__amdgpu_buffer_rsrc_t rs, uint32_t off) {
  return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
}
__device__ __forceinline__ void cl_bst(__amdgpu_buffer_rsrc_t rs, uint32_t off, const uint4& v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(cl_u32x4, v), rs, off, 0, 0);
}
// descriptor of sample n of an epilogue operand (whole-allocation extent: stores are in range by construction)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t cl_rs_sample(const rfx_cl_tensor& t, int n) {
  return cl_rsrc(reinterpret_cast<const uint16_t*>(t.p) + (int64_t)n * t.ns + t.c0, 0x7ffffff0u);
}

template <int MODE, int RW, int NT, int WM, int NTC, int KS, int DA, int DB, bool HALO>
__global__ __launch_bounds__(512, 4) void cl_conv_kernel(const ClConvK g) {
  using Cfg = ClConvCfg<RW, NT, WM, NTC, KS, DA, DB, HALO>;
  constexpr int WN = Cfg::WN, BM = Cfg::BM, MT = Cfg::MT;
  extern __shared__ __attribute__((aligned(16))) unsigned char cl_smem[];
  const rfx_cl_conv_desc& d = g.d;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;

  // ---- tile of this workgroup: position tiles sharing input rows sit on one XCD, the row groups of one position tile are adjacent
  const int bid = blockIdx.x, xcd = bid & 7, q = bid >> 3;
  const int mg = q % g.MG;
  const int pt = xcd * g.chunk + q / g.MG;
  if (pt >= g.ptiles) return;
  const int n = pt / (d.OA * g.tpr);
  const int rem = pt - n * d.OA * g.tpr;
  const int oa = rem / g.tpr, b0 = (rem - oa * g.tpr) * 256;
  constexpr int mode = MODE;                            // the epilogue is compiled per mode: no run-time dispatch, no dead operands

  // ---- bias of this row group in GEMM-row order (the vector is in the layer's own channel order: GLU rows are interleaved
  // (a_c, b_c), merged rows are (phase, channel)); read back as 16-byte groups in the epilogue
  float* bias_lds = reinterpret_cast<float*>(cl_smem + Cfg::BIAS_OFF);
  if (tid < BM) {
    const int m = mg * BM + tid;
    float bv = 0.f;
    if (d.bias != nullptr && m < d.M) {
      const int bi = mode == RFX_CL_GLU ? (m & 1) * (d.M >> 1) + (m >> 1) : ((d.Co > 0 && d.Co < d.M) ? m % d.Co : m);   // folded / merged rows: (phase, channel)
      bv = d.bias[bi];
    }
    bias_lds[tid] = bv;
  }

  // ---- valid row taps: ia = oa * SA + da0 + r * da_step in [0, IA)
  int r_lo = 0, r_hi = d.NTR;
  {
    const int base = oa * d.SA + d.da0;
    while (r_lo < r_hi && (unsigned)(base + r_lo * d.da_step) >= (unsigned)d.IA) ++r_lo;
    while (r_hi > r_lo && (unsigned)(base + (r_hi - 1) * d.da_step) >= (unsigned)d.IA) --r_hi;
  }
  const int nu = (r_hi - r_lo) * d.NCH;                     // unit u = r * NCH + c

  // ---- DMA sources
  const __amdgpu_buffer_rsrc_t rs_in = cl_rsrc(reinterpret_cast<const uint16_t*>(d.in.p) + (int64_t)n * d.in.ns + d.in.c0, g.in_bytes);
  const __amdgpu_buffer_rsrc_t rs_a = cl_rsrc(d.apack, 0x7ffffff0u);
  // Loader roles: waves 0..3 fetch the B slabs (each two 32-position blocks of every plane + the halo piece), waves 4..7 the A
  // blocks.  A wave's vmcnt then counts ONE stream, so the B stream can run DB - 1 units ahead (HBM latency) while the A stream
  // (L2-resident weights) runs DA - 1 ahead, without the in-order counter tying the two depths together.
  const bool bload = wave < 4;
  const int lw = wave & 3;
  // B core: lane -> (position 32 blk + (lane >> 1), stored half lane & 1); the stored half of slot s holds channel half e ^ bit3(s)
  int32_t core_lane[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int blk = 2 * lw + j;
    const int cslot = Cfg::CORE0 + 32 * blk + (lane >> 1);
    core_lane[j] = ((32 * blk + (lane >> 1)) * d.in.bs + 8 * ((lane & 1) ^ ((cslot >> 3) & 1))) * 2;
  }
  // halo: lanes 0..31 -> slots 0..15 (left 8, right 8)
  int32_t halo_lane = 0;
  bool halo_ok = false;
  if (HALO) {
    const int hs = (lane & 31) >> 1;
    const int p = hs < 8 ? hs - 8 : 256 + hs - 8;
    halo_lane = (p * d.in.bs + 8 * ((lane & 1) ^ ((hs >> 3) & 1))) * 2;
    halo_ok = d.wrapb ? true : ((unsigned)(b0 + p) < (unsigned)d.IB);
  }
  unsigned char* const ring_b = cl_smem + DA * Cfg::A_UNIT;

  // issue-side state of this wave's stream: next unit to fetch, its ring slot, and the running source offset (no division per unit)
  int iu = 0, islot = 0;
  int ic = 0;                                                                   // chunk index of unit iu within its row tap
  int32_t ubase = (int32_t)(((int64_t)(oa * d.SA + d.da0 + r_lo * d.da_step) * d.in.as + (int64_t)b0 * d.in.bs) * 2);
  const int32_t row_step = (int32_t)((int64_t)d.da_step * d.in.as * 2) - d.NCH * 32 * KS;
  uint32_t abase = (uint32_t)((r_lo * d.NCH * g.MG + mg) * Cfg::A_KB) * 1024u + lane * 16;
  const uint32_t a_step = (uint32_t)(g.MG * Cfg::A_KB) * 1024u;

  constexpr int dbg = RFX_CLC_DBG_BUILD;
  auto issue_next = [&]() {
    if (dbg & 1) { ++iu; return; }
    if (bload) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        unsigned char* plane = ring_b + islot * Cfg::B_UNIT + ks * Cfg::PLANE;
#pragma unroll
        for (int j = 0; j < 2; ++j)
          cl_glds16(rs_in, plane + (Cfg::CORE0 + 32 * (2 * lw + j)) * 32, (uint32_t)(ubase + core_lane[j] + 32 * ks));
        if (HALO) {
          const int32_t o = ubase + halo_lane + 32 * ks;
          const uint32_t vo = (halo_ok && o >= 0) ? (uint32_t)o : CL_OOB;
          if (lane < 32) cl_glds16(rs_in, plane, vo);            // all four loader waves: same bytes, same place
        }
      }
      ubase += 32 * KS;
      if (++ic == d.NCH) { ic = 0; ubase += row_step; }
      islot = islot + 1 == DB ? 0 : islot + 1;
    } else {
      unsigned char* buf = cl_smem + islot * Cfg::A_UNIT;
#pragma unroll
      for (int i = 0; i < Cfg::PA; ++i) {
        int blk = lw + 4 * i;
        blk = blk < Cfg::A_KB ? blk : Cfg::A_KB - 1;             // surplus pieces re-fetch the last block (same bytes, same place)
        cl_glds16(rs_a, buf + blk * 1024, abase + blk * 1024);
      }
      abase += a_step;
      islot = islot + 1 == DA ? 0 : islot + 1;
    }
    ++iu;
  };

  f32x16 acc[RW][NT];
#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;

  // B fragment addresses of this lane: tile nt, column tap t (launch-uniform taps)
  uint32_t baddr[NT][NTC];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int k = 0; k < NTC; ++k) {
      const int p = 32 * (wn * NT + t) + l31 + d.db0 + k * d.db_step;
      const int s = HALO ? cl_slot_halo(p) : p;
      baddr[t][k] = (uint32_t)(DA * Cfg::A_UNIT + s * 32 + ((h ^ ((s >> 3) & 1)) << 4));
    }
  const uint32_t afrag = (uint32_t)(wm * RW * 1024 + lane * 16);

  auto compute = [&](int sa, int sb) {
    const unsigned char* abuf = cl_smem + sa * Cfg::A_UNIT + afrag;
    const unsigned char* bbuf = cl_smem + sb * Cfg::B_UNIT;
#pragma unroll
    for (int k = 0; k < NTC; ++k)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        cl_bf16x8 bf[NT], af[RW];
        if (dbg & 2) {
#pragma unroll
          for (int t = 0; t < NT; ++t) bf[t] = __builtin_bit_cast(cl_bf16x8, make_uint4(baddr[t][k], sa, sb, lane));
#pragma unroll
          for (int i = 0; i < RW; ++i) af[i] = __builtin_bit_cast(cl_bf16x8, make_uint4(afrag, sa + i, sb, lane));
        } else {
#pragma unroll
        for (int t = 0; t < NT; ++t)
          bf[t] = __builtin_bit_cast(cl_bf16x8, *reinterpret_cast<const uint4*>(bbuf + ks * Cfg::PLANE + baddr[t][k]));
#pragma unroll
        for (int i = 0; i < RW; ++i)
          af[i] = __builtin_bit_cast(cl_bf16x8, *reinterpret_cast<const uint4*>(abuf + ((k * KS + ks) * MT + i) * 1024));
        }
        if (dbg & 4) {
#pragma unroll
          for (int i = 0; i < RW; ++i)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[i][t][0] += __builtin_bit_cast(float, (uint32_t)af[i][0] ^ (uint32_t)bf[t][1]);
        } else {
#pragma unroll
        for (int i = 0; i < RW; ++i)
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[t], acc[i][t], 0, 0, 0);
        }
      }
  };

  // ---- pipeline over the valid units.  At the wait of iteration i this wave has issued units 0 .. i + D - 2 of its stream
  // (D = its ring depth): unit i has landed when at most min(left, D - 2) later units' pieces are outstanding.
  if (nu > 0) {
    const int ahead = bload ? DB - 1 : DA - 1;
    for (int i = 0; i < ahead; ++i)
      if (iu < nu) issue_next();
    int sa = 0, sb = 0;
    for (int i = 0; i < nu; ++i) {
      const int left = nu - 1 - i;
      if (bload) {
        if (left >= DB - 2) cl_wait_vm_c<(DB - 2) * Cfg::PB>(); else cl_wait_vm(left * Cfg::PB);
      } else {
        if (left >= DA - 2) cl_wait_vm_c<(DA - 2) * Cfg::PA>(); else cl_wait_vm(left * Cfg::PA);
      }
      if (!(dbg & 16)) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (iu < nu) issue_next();                     // into the slot whose unit every wave finished before this barrier
      __builtin_amdgcn_sched_barrier(0);
      compute(sa, sb);
      sa = sa + 1 == DA ? 0 : sa + 1;
      sb = sb + 1 == DB ? 0 : sb + 1;
    }
  }
  __builtin_amdgcn_s_barrier();                 // every wave is done reading the rings: their memory becomes the transpose tiles

  if (dbg & 8) {
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < RW; ++i)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[i][t][r];
    if (sum == 12345.678f) reinterpret_cast<float*>(d.out0.p ? d.out0.p : d.out1.p)[tid] = sum;
    return;
  }
  // ---- epilogue
  unsigned char* ez = cl_smem + wave * Cfg::EPI_WAVE;            // [32 positions][32 RW rows] bf16, row stride EPI_RS
  unsigned char* ey = ez + 32 * Cfg::EPI_RS;                     // GLU: [32][16 RW] bf16, row stride EPI_RSY
  const int mrow0 = mg * BM + wm * 32 * RW;                      // first GEMM row of this wave
  constexpr int CH = 4 * RW;                                     // 16-byte groups per position in the z tile
  constexpr int CHY = 2 * RW;
  const int Cglu = d.M >> 1;                                     // GLU: channels of one half
  const float* bl = bias_lds + wm * 32 * RW + 4 * h;

  const __amdgpu_buffer_rsrc_t rs0 = cl_rs_sample(d.out0, n), rs1 = cl_rs_sample(d.out1, n);
  const __amdgpu_buffer_rsrc_t rsx = cl_rs_sample(d.aux0, n), rsr = cl_rs_sample(d.res, n);
  const bool has0 = d.out0.p != nullptr, has_aux = d.aux0.p != nullptr, has_res = d.res.p != nullptr;
  const bool has1 = d.out1.p != nullptr;
  // GLU + per-row vector (the frequency embedding of Hybrid Demucs, added to the first encoder layer's output): the values of this
  // lane's channels, position-independent
  float ra[RW][4][2];
#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      ra[i][gq][0] = ra[i][gq][1] = 0.f;
      if (mode == RFX_CL_GLU && d.rowadd != nullptr) {
        const int c = (mrow0 >> 1) + 16 * i + 4 * gq + 2 * h;
        if (c < Cglu) {
          ra[i][gq][0] = d.rowadd[(int64_t)oa * Cglu + c];
          ra[i][gq][1] = d.rowadd[(int64_t)oa * Cglu + c + 1];
        }
      }
    }

#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (mode == RFX_CL_STORE_CM) {
      // channel-major fp32 straight from the accumulators: a register is one GEMM row = (sub-row / sub-position, channel), the lanes
      // are 32 consecutive positions -- 128-byte runs.  The last decoder layer's transposed convolution (C -> 1 or 2 channels).
      const int bp = b0 + 32 * (wn * NT + t) + l31;
      float* cm = reinterpret_cast<float*>(d.cm_out) + (int64_t)n * d.cm_ns;
      const int nsub = d.M / d.Co;
#pragma unroll
      for (int i = 0; i < RW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h, m = mrow0 + row;
          if (m < d.M) {
            const int psi = m / d.Co, co = m - psi * d.Co;
            const int orow = d.cm_fold ? oa : oa * d.G + psi + d.g_off;
            const int64_t pos = d.cm_fold ? (int64_t)bp * nsub + psi : bp;
            if (d.cm_fold || (unsigned)orow < (unsigned)d.OAo)
              cm[(int64_t)co * d.cm_cs + (int64_t)orow * d.cm_as + pos] = acc[i][t][r] + bias_lds[wm * 32 * RW + row];
          }
        }
      continue;
    }
    // registers -> transpose tile
#pragma unroll
    for (int i = 0; i < RW; ++i)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const f32x4 bq = *reinterpret_cast<const f32x4*>(bl + 32 * i + 8 * gq);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][t][4 * gq + e] + bq[e];
        if (mode == RFX_CL_GLU) {
          // rows (a_c, b_c, a_c+1, b_c+1), c = (32 i + 8 gq + 4 h) / 2 within the wave: z tile holds [a part | b part]
          const int cl = 16 * i + 4 * gq + 2 * h;
          *reinterpret_cast<uint32_t*>(ez + l31 * Cfg::EPI_RS + cl * 2) = rfx_cvt_pk_bf16(v[0], v[2]);
          *reinterpret_cast<uint32_t*>(ez + l31 * Cfg::EPI_RS + (16 * RW + cl) * 2) = rfx_cvt_pk_bf16(v[1], v[3]);
          *reinterpret_cast<uint32_t*>(ey + l31 * Cfg::EPI_RSY + cl * 2) =
              rfx_cvt_pk_bf16(v[0] * rfx_sigmoid(v[1]) + ra[i][gq][0], v[2] * rfx_sigmoid(v[3]) + ra[i][gq][1]);
        } else {
          *reinterpret_cast<uint2*>(ez + l31 * Cfg::EPI_RS + (32 * i + 8 * gq + 4 * h) * 2) =
              make_uint2(rfx_cvt_pk_bf16(v[0], v[1]), rfx_cvt_pk_bf16(v[2], v[3]));
        }
      }
    CL_LGKM0();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_wave_barrier();
    const int bpos0 = b0 + 32 * (wn * NT + t);
    if (mode == RFX_CL_GLU) {
      // z: the wave's a-part covers natural channels [mrow0 / 2, +16 RW), its b-part Cglu + the same
      const int ca = mrow0 >> 1;
      if (has0) {
        const uint32_t o0 = (uint32_t)(((int64_t)oa * d.out0.as + (int64_t)bpos0 * d.out0.bs + ca) * 2);
#pragma unroll
        for (int it = 0; it < (32 * CH + 63) / 64; ++it) {
          const int f = lane + 64 * it;
          const int pos = f / CH, cg = f - pos * CH;                // cg < 2 RW: a part, else b part
          const int part = cg >= CHY ? 1 : 0, cc = cg - part * CHY;
          if (f < 32 * CH && ca + 8 * cc < Cglu) {
            const uint2 lo = *reinterpret_cast<const uint2*>(ez + pos * Cfg::EPI_RS + cg * 16);
            const uint2 hi = *reinterpret_cast<const uint2*>(ez + pos * Cfg::EPI_RS + cg * 16 + 8);
            cl_bst(rs0, o0 + (uint32_t)((pos * d.out0.bs + part * Cglu + 8 * cc) * 2), make_uint4(lo.x, lo.y, hi.x, hi.y));
          }
        }
      }
      const uint32_t o1 = (uint32_t)(((int64_t)oa * d.out1.as + (int64_t)bpos0 * d.out1.bs + ca) * 2);
#pragma unroll
      for (int it = 0; it < (32 * CHY + 63) / 64; ++it) {
        const int f = lane + 64 * it;
        const int pos = f / CHY, cg = f - pos * CHY;
        if (f < 32 * CHY && ca + 8 * cg < Cglu) {
          const uint2 lo = *reinterpret_cast<const uint2*>(ey + pos * Cfg::EPI_RSY + cg * 16);
          const uint2 hi = *reinterpret_cast<const uint2*>(ey + pos * Cfg::EPI_RSY + cg * 16 + 8);
          cl_bst(rs1, o1 + (uint32_t)((pos * d.out1.bs + 8 * cg) * 2), make_uint4(lo.x, lo.y, hi.x, hi.y));
        }
      }
    } else {
#pragma unroll
      for (int it = 0; it < (32 * CH + 63) / 64; ++it) {
        const int f = lane + 64 * it;
        const int pos = f / CH, cg = f - pos * CH;
        const int m = mrow0 + 8 * cg;
        int orow = oa, ch = m;
        bool ok = f < 32 * CH && m < d.M;
        if (d.G > 1) {                                           // merged phases: G <= 4
          const int psi = (m >= d.Co ? 1 : 0) + (m >= 2 * d.Co ? 1 : 0) + (m >= 3 * d.Co ? 1 : 0);
          ch = m - psi * d.Co;
          orow = oa * d.G + psi + d.g_off;
          ok = ok && (unsigned)orow < (unsigned)d.OAo;
        }
        if (!ok) continue;
        const uint2 lo = *reinterpret_cast<const uint2*>(ez + pos * Cfg::EPI_RS + cg * 16);
        const uint2 hi = *reinterpret_cast<const uint2*>(ez + pos * Cfg::EPI_RS + cg * 16 + 8);
        uint4 raw = make_uint4(lo.x, lo.y, hi.x, hi.y);
        const int bp = bpos0 + pos;
        float v[8];
        if (has_res) {
          const uint4 rr = cl_bld(rsr, (uint32_t)(((int64_t)orow * d.res.as + (int64_t)bp * d.res.bs + ch) * 2));
          float w[8];
          cl_unpack8(raw, v); cl_unpack8(rr, w);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += w[e];
          raw = cl_pack8(v);
        }
        const uint32_t off0 = (uint32_t)(((int64_t)orow * d.out0.as + (int64_t)bp * d.out0.bs + ch) * 2);
        if (mode == RFX_CL_STORE) {
          cl_bst(rs0, off0, raw);
          continue;
        }
        cl_unpack8(raw, v);                                      // the rounded values: what a later pass reads back
        if (mode == RFX_CL_GELU || mode == RFX_CL_DGELU) {
          if (has0) cl_bst(rs0, off0, raw);
          float w[8];
          if (has_aux) cl_unpack8(cl_bld(rsx, (uint32_t)(((int64_t)orow * d.aux0.as + (int64_t)bp * d.aux0.bs + ch) * 2)), w);
          float o[8];
          if (mode == RFX_CL_GELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = rfx_gelu(v[e]) + (has_aux ? w[e] : 0.f);
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = v[e] * rfx_gelu_grad(w[e]);
          }
          cl_bst(rs1, (uint32_t)(((int64_t)orow * d.out1.as + (int64_t)bp * d.out1.bs + ch) * 2), cl_pack8(o));
        } else {                                                 // RFX_CL_DGLU: aux0 = stored [a | b] of the forward pass, Co channels each
          const uint32_t ax = (uint32_t)(((int64_t)orow * d.aux0.as + (int64_t)bp * d.aux0.bs + ch) * 2);
          const int half = d.aux0.bs >> 1;
          float a[8], b[8], ga[8], gb[8];
          if (has1) cl_bst(rs1, (uint32_t)(((int64_t)orow * d.out1.as + (int64_t)bp * d.out1.bs + ch) * 2), raw);   // the summed gradient itself
          cl_unpack8(cl_bld(rsx, ax), a);
          cl_unpack8(cl_bld(rsx, ax + half * 2), b);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float s = rfx_sigmoid(b[e]);
            ga[e] = v[e] * s;
            gb[e] = v[e] * a[e] * s * (1.f - s);
          }
          cl_bst(rs0, off0, cl_pack8(ga));
          cl_bst(rs0, off0 + half * 2, cl_pack8(gb));
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    CL_LGKM0();
  }
}

template <int MODE, int RW, int NT, int WM, int NTC, int KS, int DA, int DB, bool HALO>
static int cl_conv_launch(const ClConvK& k, dim3 grid, hipStream_t s) {
  using Cfg = ClConvCfg<RW, NT, WM, NTC, KS, DA, DB, HALO>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&cl_conv_kernel<MODE, RW, NT, WM, NTC, KS, DA, DB, HALO>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS) != hipSuccess)
      return -3;
    attr_done = true;
  }
  hipLaunchKernelGGL((cl_conv_kernel<MODE, RW, NT, WM, NTC, KS, DA, DB, HALO>), grid, dim3(512), Cfg::LDS, s, k);
  RFX_CHECK_LAUNCH();
  return 0;
}

// tap geometry -> instantiation.  Ring depths fill the LDS the epilogue's transpose tiles need anyway (8 x 9.5 KiB for 96-row
// waves): the B stream runs 2..5 units ahead, the A stream one.
template <int MODE, int RW, int NT, int WM>
static int cl_conv_pick(const ClConvK& k, dim3 grid, hipStream_t s) {
  const rfx_cl_conv_desc& d = k.d;
  const bool halo = d.NTC > 1 || d.db0 != 0;
  constexpr int MT = RW * WM;
  if (d.NTC == 3 && d.KS == 1 && halo) return cl_conv_launch<MODE, RW, NT, WM, 3, 1, 2, (MT >= 6 ? 4 : 6), true>(k, grid, s);
  if (d.NTC == 1 && d.KS == 2 && !halo) return cl_conv_launch<MODE, RW, NT, WM, 1, 2, 2, (MT >= 6 ? 3 : 4), false>(k, grid, s);
  if (d.NTC == 1 && d.KS == 1 && !halo) return cl_conv_launch<MODE, RW, NT, WM, 1, 1, 2, 6, false>(k, grid, s);
  return -1;
}

// tile height -> instantiation (the channel-major store exists for 32-row tiles only)
template <int MODE>
static int cl_conv_dispatch(const ClConvK& k, dim3 grid, hipStream_t s) {
  if constexpr (MODE == RFX_CL_STORE_CM) {
    return k.d.BM == 32 ? cl_conv_pick<MODE, 1, 1, 1>(k, grid, s) : -1;
  } else {
    switch (k.d.BM) {
      case 192: return cl_conv_pick<MODE, 3, 2, 2>(k, grid, s);
      case 96: return cl_conv_pick<MODE, 3, 1, 1>(k, grid, s);
      case 64: return cl_conv_pick<MODE, 2, 1, 1>(k, grid, s);
      case 32: return cl_conv_pick<MODE, 1, 1, 1>(k, grid, s);
    }
    return -1;
  }
}

// one per epilogue mode (csrc/cl_conv_m_*.hip)
int cl_conv_mode_store(const ClConvK& k, dim3 grid, hipStream_t s);
int cl_conv_mode_gelu(const ClConvK& k, dim3 grid, hipStream_t s);
int cl_conv_mode_glu(const ClConvK& k, dim3 grid, hipStream_t s);
int cl_conv_mode_dgelu(const ClConvK& k, dim3 grid, hipStream_t s);
int cl_conv_mode_dglu(const ClConvK& k, dim3 grid, hipStream_t s);
int cl_conv_mode_cm(const ClConvK& k, dim3 grid, hipStream_t s);
