// Halo-tile implicit GEMM (gemm_halo.h): instantiations for fp32- and bf16-stored operands, bf16 arithmetic.
#include "gemm_halo.h"

int rfx_launch_gemm_halo(const FwdArgs& g0, hipStream_t s) {
  FwdArgs g = g0;
  const rfx_gemm_desc& d = g.d;
  const int r = rfx_halo_pick_r(d);
  if (r <= 0 || !rfx_halo_geo_ok(d)) return -1;
  const int64_t work = (int64_t)d.N * d.OA * (d.OB / RFX_HALO_TW);
  const int64_t nblk = ((work + 7) / 8) * 8 * (d.Mpad / (32 * r));
  if (nblk > 0x7fffffff) return -1;
  // neighbouring rows share two of their three input rows: contiguous runs of work items per XCD (its L2 serves the overlap)
  g.xcd_chunk = d.OA > 1 ? (int)((work + 7) / 8) : 0;
  dim3 grid((unsigned)nblk);
  return d.in_bf16 ? rfx_launch_halo_variant<1>(g, r, grid, s) : rfx_launch_halo_variant<0>(g, r, grid, s);
}
