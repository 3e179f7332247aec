// Weight gradient, bf16-operand instantiations.
#include "gemm_wgrad.h"

int rfx_launch_wgrad_bf16(const WgradArgs& w, int shape, dim3 grid, hipStream_t s) { return rfx_launch_wgrad_bf<2>(w, shape, grid, s); }
int rfx_launch_wgrad_wide_bf16(const WgradArgs& w, int shape, dim3 grid, hipStream_t s) { return rfx_launch_wgrad_wide<2>(w, shape, grid, s); }
