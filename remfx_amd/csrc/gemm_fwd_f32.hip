// Tiled gather-GEMM forward kernel, channel-major table, exact fp32 (v_mfma_f32_32x32x2_f32) instantiations.
#include "gemm_fwd.h"

int rfx_launch_gemm_fwd_f32(const FwdArgs& g, int r, dim3 grid, hipStream_t s) {
  switch (r) {
    case 1: hipLaunchKernelGGL((gemm_fwd_kernel<1>), grid, dim3(256), 0, s, g); break;
    case 2: hipLaunchKernelGGL((gemm_fwd_kernel<2>), grid, dim3(256), 0, s, g); break;
    case 3: hipLaunchKernelGGL((gemm_fwd_kernel<3>), grid, dim3(256), 0, s, g); break;
    default: hipLaunchKernelGGL((gemm_fwd_kernel<4>), grid, dim3(256), 0, s, g); break;
  }
  RFX_CHECK_LAUNCH();
  return 0;
}
