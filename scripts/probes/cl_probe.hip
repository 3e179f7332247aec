// Hardware-semantics probe for the channels-last kernels (gfx950): ds_read_b64_tr_b16 lane mapping, buffer_load ... lds with
// out-of-range / exec-masked lanes.  Build: hipcc --offload-arch=gfx950 -O2 cl_probe.hip -o cl_probe.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))

// P1: LDS u16[i] = i; lane l reads tr16 at byte address addr[l]; out[l*4+e]
__global__ void p1(const int* addr, uint16_t* out) {
  const int lane = threadIdx.x;
  uint16_t* s = (uint16_t*)smem;
  for (int i = lane; i < 4096; i += 64) s[i] = (uint16_t)i;
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smem + addr[lane]));
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (uint16_t)v[e];
}
// P2: LDS prefilled with 0xAAAA; buffer_load_dwordx4 ... lds with voffset[l] (some out of range, some lanes masked off)
__global__ void p2(const uint32_t* src, int nbytes, const uint32_t* voff, const int* active, uint32_t* out) {
  const int lane = threadIdx.x;
  uint32_t* s = (uint32_t*)smem;
  for (int i = lane; i < 1024; i += 64) s[i] = 0xAAAAAAAAu;
  __syncthreads();
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  if (active[lane])
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDSP(smem + 1024), 16, voff[lane], 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  for (int i = lane; i < 1024; i += 64) out[i] = s[i];
}
int main() {
  int haddr[64]; uint16_t hout[256];
  int *daddr; uint16_t* dout;
  hipMalloc(&daddr, sizeof(haddr)); hipMalloc(&dout, sizeof(hout));
  for (int variant = 0; variant < 3; ++variant) {
    for (int l = 0; l < 64; ++l) {
      if (variant == 0) haddr[l] = l * 8;                          // natural: lane l -> 8-byte chunk l
      if (variant == 1) haddr[l] = (l & 15) * 64 + (l >> 4) * 8;   // 16 rows of 64 B per group, groups 8 B apart
      if (variant == 2) haddr[l] = ((l >> 2) & 3) * 192 + (l & 3) * 8 + (l >> 4) * 32 + 2048;  // rows of 192 B: chunk (row=(l>>2)&3, col4=(l&3))
    }
    hipMemcpy(daddr, haddr, sizeof(haddr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(p1, dim3(1), dim3(64), 8192, 0, daddr, dout);
    hipMemcpy(hout, dout, sizeof(hout), hipMemcpyDeviceToHost);
    printf("P1 variant %d (value = u16 index in LDS):\n", variant);
    for (int l = 0; l < 64; ++l) printf("  lane %2d addr %4d -> %4d %4d %4d %4d\n", l, haddr[l], hout[l*4], hout[l*4+1], hout[l*4+2], hout[l*4+3]);
  }
  // P2
  std::vector<uint32_t> hsrc(1024); for (int i = 0; i < 1024; ++i) hsrc[i] = 0x10000u + i;
  uint32_t hv[64]; int hact[64];
  for (int l = 0; l < 64; ++l) { hv[l] = l * 16; hact[l] = 1; }
  hv[3] = 0x80000000u; hv[5] = 4096;   // far OOB, just past the end (nbytes = 2048 below -> lanes >= 128*.. also OOB)
  hact[7] = 0; hact[40] = 0;
  uint32_t *dsrc, *dv, *do2; int* dact;
  hipMalloc(&dsrc, 4096); hipMalloc(&dv, 256); hipMalloc(&dact, 256); hipMalloc(&do2, 4096);
  hipMemcpy(dsrc, hsrc.data(), 4096, hipMemcpyHostToDevice); hipMemcpy(dv, hv, 256, hipMemcpyHostToDevice);
  hipMemcpy(dact, hact, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(p2, dim3(1), dim3(64), 8192, 0, dsrc, 800 /* bytes: lanes 50.. are out of range */, dv, dact, do2);
  std::vector<uint32_t> ho(1024); hipMemcpy(ho.data(), do2, 4096, hipMemcpyDeviceToHost);
  printf("P2 (LDS dwords 256.. = landing zone; src[i]=0x10000+i; num_records=800 B):\n");
  for (int l = 0; l < 64; ++l) printf("  lane %2d voff %08x act %d -> %08x %08x %08x %08x\n", l, hv[l], hact[l], ho[256+l*4], ho[256+l*4+1], ho[256+l*4+2], ho[256+l*4+3]);
  printf("  guard before %08x after %08x\n", ho[255], ho[256 + 256]);
  return 0;
}
