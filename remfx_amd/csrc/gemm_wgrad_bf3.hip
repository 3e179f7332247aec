// Weight gradient, split-bf16x3 instantiations.
#include "gemm_wgrad.h"

int rfx_launch_wgrad_bf3(const WgradArgs& w, int shape, dim3 grid, hipStream_t s) { return rfx_launch_wgrad_bf<1>(w, shape, grid, s); }
int rfx_launch_wgrad_wide_bf3(const WgradArgs& w, int shape, dim3 grid, hipStream_t s) { return rfx_launch_wgrad_wide<1>(w, shape, grid, s); }
