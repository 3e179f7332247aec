// Implicit-GEMM convolution on channels-last bf16 operands (rfx_cl_conv): the frequency branch of Hybrid Demucs in the bf16
// arithmetic mode -- torchaudio HDemucs `_HEncLayer.conv / rewrite`, `_HDecLayer.rewrite / conv_tr` and their input gradients,
// reached from remfx/models.py:308,317.
//
// Structure (one workgroup = 8 waves = BM GEMM rows x 256 consecutive positions of one (n, row)):
//   * the reduction is cut into UNITS = (row tap r, chunk c of 16 KS channels).  A unit's B slab -- 256 (+ 2 x 8 halo) positions x
//     16 KS channels, KS planes of [position][32 bytes] -- and its packed A block (NTC x KS x BM/32 MFMA fragments of 1 KiB) are
//     fetched by `buffer_load ... lds` DMA: no VGPRs, no conversion, no per-element address work; positions outside the tensor take
//     the out-of-range offset and land as zeros.
//   * units run through two LDS rings, B slabs DB deep and A blocks DA = 2 deep (weights are L2-resident).  Waves 0..3 issue the B
//     pieces, waves 4..7 the A pieces, so each wave's in-order vmcnt counts one stream.  Per unit: wait (counted vmcnt) for unit u,
//     ONE workgroup barrier, issue the units that far ahead, then NTC x KS x RW x NT MFMAs per wave.  A unit's column taps are
//     served from the same slab at shifted positions, so a 3x3 layer reads each input value once per row tap from L2 and never per
//     column tap.
//   * B slab layout: slot s = position, 32 bytes; the two 16-byte halves of a slot are swapped when bit 3 of s is set, which makes
//     the ds_read_b128 of an MFMA B fragment (32 consecutive slots, one half) conflict-free in its 16-lane service groups for any
//     column shift.  The DMA writes lane-linearly, so the swap is applied to the SOURCE address (guide rule 21).
//   * A fragments are stored in fragment order by rfx_cl_pack: reading one is ds_read_b128 at lane * 16.
//   * epilogue: bias (staged once in LDS in GEMM-row order) and the in-register part of the mode (GLU on interleaved rows),
//     rounding to bf16, a per-wave LDS transpose ([32 positions][rows], 8-byte pad), then 16-byte channel groups go out as full
//     lines through buffer stores together with the elementwise tail of the mode (GELU + skip add, GELU / GLU backward against
//     the stored forward tensor, residual-gradient add).
// The first version spent 2200 scalar + vector instructions per wave and 256-position tile around 81 MFMAs (SQ counters, r05):
// per-unit integer division, a jump table for the counted wait, 48 scalar bias loads, 64-bit address arithmetic per stored group.
#include "cl_conv.h"

static bool cl_fits32(const rfx_cl_tensor& t, int rows, int IB) {
  if (!t.p) return true;
  return ((int64_t)rows * t.as + (int64_t)IB * t.bs) * 2 < 0x7fffffffLL && t.bs % 8 == 0 && t.c0 % 8 == 0;
}

extern "C" int rfx_cl_conv(const rfx_cl_conv_desc* dp, void* stream) {
  if (!dp || !dp->in.p || !dp->apack) return -1;
  const rfx_cl_conv_desc& d = *dp;
  if (d.N <= 0 || d.OA <= 0 || d.OB <= 0 || d.OB % 256 || d.OB != d.IB || d.M <= 0 || d.NTR <= 0 || d.NTR > 16 || d.NCH <= 0) return -1;
  if (d.in.bs % 8 || d.in.c0 % 8 || (d.NCH * 16 * d.KS + d.in.c0) > d.in.bs) return -1;
  const bool cm = d.mode == RFX_CL_STORE_CM;
  if (d.G < 1 || d.G > 4 || (d.G > 1 && ((!cm && d.Co % 8) || d.Co * d.G != d.M))) return -1;
  if (d.mode < 0 || d.mode > RFX_CL_STORE_CM) return -1;
  if (cm && (!d.cm_out || d.Co <= 0 || d.M % d.Co || d.BM != 32)) return -1;
  if (d.mode == RFX_CL_GLU && (d.M % 16 || !d.out1.p || d.G != 1)) return -1;
  if ((d.mode == RFX_CL_STORE || d.mode == RFX_CL_DGLU) && !d.out0.p) return -1;
  if ((d.mode == RFX_CL_GELU || d.mode == RFX_CL_DGELU) && !d.out1.p) return -1;
  if ((d.mode == RFX_CL_DGELU || d.mode == RFX_CL_DGLU) && !d.aux0.p) return -1;
  if (d.M % 8 && !cm) return -1;
  const int orows = d.G > 1 ? d.OAo : d.OA;
  if (!cl_fits32(d.out0, orows, d.IB) || !cl_fits32(d.out1, orows, d.IB) || !cl_fits32(d.aux0, orows, d.IB) || !cl_fits32(d.res, orows, d.IB))
    return -1;
  const int64_t ext = ((int64_t)(d.IA - 1) * d.in.as + (int64_t)d.IB * d.in.bs - d.in.c0) * 2;
  if (ext <= 0 || ext >= 0x7fffffffLL) return -1;
  ClConvK k;
  k.d = d;
  k.tpr = d.OB / 256;
  k.ptiles = d.N * d.OA * k.tpr;
  k.chunk = (k.ptiles + 7) / 8;
  k.MG = (d.M + d.BM - 1) / d.BM;
  k.in_bytes = (uint32_t)ext;

  const dim3 grid((unsigned)(8 * k.chunk * k.MG));
  hipStream_t s = (hipStream_t)stream;
  switch (d.mode) {
    case RFX_CL_STORE: return cl_conv_mode_store(k, grid, s);
    case RFX_CL_GELU: return cl_conv_mode_gelu(k, grid, s);
    case RFX_CL_GLU: return cl_conv_mode_glu(k, grid, s);
    case RFX_CL_DGELU: return cl_conv_mode_dgelu(k, grid, s);
    case RFX_CL_DGLU: return cl_conv_mode_dglu(k, grid, s);
    case RFX_CL_STORE_CM: return cl_conv_mode_cm(k, grid, s);
  }
  return -1;
}

// ---- weight packing -----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cl_pack_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx, int64_t n,
                                                      uint16_t* __restrict__ dst) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int32_t j = idx[i];
    dst[i] = (uint16_t)rfx_bf16_bits(j < 0 ? 0.f : src[j]);
  }
}
extern "C" int rfx_cl_pack(const float* src, const int32_t* idx, int64_t n, void* dst, void* stream) {
  if (!src || !idx || !dst || n < 0) return -1;
  if (n == 0) return 0;
  const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(cl_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, idx, n, reinterpret_cast<uint16_t*>(dst));
  RFX_CHECK_LAUNCH();
  return 0;
}
