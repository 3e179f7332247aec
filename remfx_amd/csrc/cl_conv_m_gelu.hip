// rfx_cl_conv, epilogue mode RFX_CL_GELU (kernel template: csrc/cl_conv.h)
#include "cl_conv.h"

int cl_conv_mode_gelu(const ClConvK& k, dim3 grid, hipStream_t s) { return cl_conv_dispatch<RFX_CL_GELU>(k, grid, s); }
