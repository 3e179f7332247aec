// Tap-major gather-GEMM forward kernel, bf16 operands (1 x v_mfma_f32_32x32x16_bf16 per 16-deep K step) instantiations.
#include "gemm_tap.h"

int rfx_launch_gemm_fwd_bf16(const FwdArgs& g, int r, dim3 grid, hipStream_t s) { return rfx_launch_gemm_tap<2>(g, r, grid, s); }
