// Tap-major gather-GEMM forward kernel, bf16 arithmetic, gathered operand STORED as bf16 (rfx_gemm_desc.in_bf16 == 1).
#include "gemm_tap.h"

int rfx_launch_gemm_tap_in16(const FwdArgs& g, int r, dim3 grid, hipStream_t s) { return rfx_launch_gemm_tap_variant<1>(g, r, grid, s); }
