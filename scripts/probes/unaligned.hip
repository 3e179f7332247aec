// probe: raw buffer loads of 8 bytes at 2-byte-aligned offsets (bf16 operand at an odd tap shift)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(const uint16_t* in, uint32_t* out, int n) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(in), 0, n * 2, 0x00020000);
  const int t = threadIdx.x;
  typedef uint32_t u2 __attribute__((ext_vector_type(2)));
  u2 v = __builtin_bit_cast(u2, __builtin_amdgcn_raw_buffer_load_b64(rs, (uint32_t)t * 2u, 0, 0));
  out[2 * t] = v.x; out[2 * t + 1] = v.y;
}
int main() {
  const int n = 256;
  uint16_t h[n]; for (int i = 0; i < n; ++i) h[i] = (uint16_t)(0x1000 + i);
  uint16_t* d; uint32_t* o; uint32_t ho[2 * 64];
  hipMalloc(&d, n * 2); hipMalloc(&o, sizeof(ho)); hipMemcpy(d, h, n * 2, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
  hipError_t e = hipDeviceSynchronize();
  hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 64; ++t) {
    const uint32_t e0 = (0x1000u + t) | ((0x1000u + t + 1) << 16), e1 = (0x1000u + t + 2) | ((0x1000u + t + 3) << 16);
    if (ho[2 * t] != e0 || ho[2 * t + 1] != e1) { if (bad < 6) printf("t=%d got %08x %08x want %08x %08x\n", t, ho[2 * t], ho[2 * t + 1], e0, e1); ++bad; }
  }
  printf("unaligned b64 buffer loads: %s (%d mismatches), err %d\n", bad ? "WRONG" : "ok", bad, (int)e);
  return 0;
}
