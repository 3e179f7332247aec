// Tap-major gather-GEMM forward kernel on a channels-last bf16 operand (rfx_gemm_desc.in_bf16 == 3): probe of the next layout,
// scripts/probes/cl_gather_probe.py, DESIGN 8.8.  Not used by the product path.
#include "gemm_tap.h"

int rfx_launch_gemm_tap_cl(const FwdArgs& g, int r, dim3 grid, hipStream_t s) { return rfx_launch_gemm_tap_variant<3>(g, r, grid, s); }
