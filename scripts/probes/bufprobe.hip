// Hardware-semantics probe (dev tool, not part of the library): raw buffer loads on gfx950.
//  1. unaligned (4-byte aligned) dwordx4 loads  2. per-dword range check of a dwordx4 that straddles num_records
//  3. is soffset part of the range check?       4. negative (wrapped) voffset
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float* base, float* out, uint32_t nrec, uint32_t soff) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, nrec, 0x00020000);
  const int l = threadIdx.x;
  // 1: lane l loads 4 floats at byte offset 4 + 16 l  (unaligned)
  u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, 4u + 16u * l, 0, 0);
  for (int i = 0; i < 4; ++i) out[l * 4 + i] = __uint_as_float(a[i]);
  // 2: straddle: offset nrec - 8 -> dwords 0,1 in range, 2,3 out
  u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, nrec - 8u, 0, 0);
  if (l == 0) for (int i = 0; i < 4; ++i) out[256 + i] = __uint_as_float(b[i]);
  // 3: voffset in range, soffset pushes beyond num_records
  uint32_t c = __builtin_amdgcn_raw_buffer_load_b32(rs, nrec - 16u, soff, 0);
  if (l == 0) out[260] = __uint_as_float(c);
  // 3b: voffset beyond, soffset = 0
  uint32_t c2 = __builtin_amdgcn_raw_buffer_load_b32(rs, nrec + 16u, 0, 0);
  if (l == 0) out[261] = __uint_as_float(c2);
  // 4: wrapped negative offset
  u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(rs, 0xfffffffcu, 0, 0);
  if (l == 0) for (int i = 0; i < 4; ++i) out[262 + i] = __uint_as_float(d[i]);
}
int main() {
  const int N = 4096;
  float *d, *o, h[N], ho[512];
  for (int i = 0; i < N; ++i) h[i] = (float)i;
  hipMalloc(&d, N * 4); hipMalloc(&o, 512 * 4);
  hipMemcpy(d, h, N * 4, hipMemcpyHostToDevice);
  hipMemset(o, 0, 512 * 4);
  const uint32_t nrec = 2048;   // bytes -> 512 floats in range
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, nrec, 64u);
  hipMemcpy(ho, o, 512 * 4, hipMemcpyDeviceToHost);
  printf("1 unaligned x4: lane0 %g %g %g %g  lane5 %g %g %g %g (expect 1 2 3 4 / 21 22 23 24)\n", ho[0], ho[1], ho[2], ho[3], ho[20], ho[21], ho[22], ho[23]);
  printf("2 straddle at nrec-8: %g %g %g %g (per-dword check -> 510 511 0 0)\n", ho[256], ho[257], ho[258], ho[259]);
  printf("3 voff=nrec-16 soff=64: %g (508+16=524 if soffset NOT range-checked, 0 if it is); voff=nrec+16: %g (expect 0)\n", ho[260], ho[261]);
  printf("4 voff=0xfffffffc x4: %g %g %g %g (all 0 expected; wrap-around would give ? 0 1 2)\n", ho[262], ho[263], ho[264], ho[265]);
  return 0;
}
