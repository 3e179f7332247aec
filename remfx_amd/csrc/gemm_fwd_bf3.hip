// Tiled gather-GEMM forward kernel, split-bf16x3 (3 x v_mfma_f32_32x32x16_bf16 per 16-deep K step) instantiations.
#include "gemm_fwd.h"

int rfx_launch_gemm_fwd_bf3(const FwdArgs& g, int r, dim3 grid, hipStream_t s) {
  switch (r) {
    case 1: hipLaunchKernelGGL((gemm_fwd_kernel<1, true>), grid, dim3(256), 0, s, g); break;
    case 2: hipLaunchKernelGGL((gemm_fwd_kernel<2, true>), grid, dim3(256), 0, s, g); break;
    case 3: hipLaunchKernelGGL((gemm_fwd_kernel<3, true>), grid, dim3(256), 0, s, g); break;
    default: hipLaunchKernelGGL((gemm_fwd_kernel<4, true>), grid, dim3(256), 0, s, g); break;
  }
  RFX_CHECK_LAUNCH();
  return 0;
}
