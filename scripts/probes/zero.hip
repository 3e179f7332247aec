// probe: zero-fill bandwidth variants
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ __launch_bounds__(256) void z1(uint4* q, int64_t n4) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
  for (int64_t i = gid; i < n4; i += stride) q[i] = make_uint4(0u, 0u, 0u, 0u);
}
// each block owns a contiguous span
__global__ __launch_bounds__(256) void z2(uint4* q, int64_t n4, int64_t per_block) {
  const int64_t b0 = (int64_t)blockIdx.x * per_block, b1 = min(b0 + per_block, n4);
  for (int64_t i = b0 + threadIdx.x; i < b1; i += 256) q[i] = make_uint4(0u, 0u, 0u, 0u);
}
__global__ __launch_bounds__(256) void z3(uint4* q, int64_t n4, int64_t per_block) {
  const int64_t b0 = (int64_t)blockIdx.x * per_block, b1 = min(b0 + per_block, n4);
  for (int64_t i = b0 + threadIdx.x; i < b1; i += 256) { typedef unsigned int u4 __attribute__((ext_vector_type(4))); __builtin_nontemporal_store(u4{0u, 0u, 0u, 0u}, reinterpret_cast<u4*>(q + i)); }
}
int main() {
  const int64_t bytes = 334ll << 20, n4 = bytes / 16;
  uint4* d; hipMalloc(&d, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto f) {
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.1f us  %6.2f TB/s\n", name, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12);
  };
  run("grid-stride 16384 blocks", [&] { hipLaunchKernelGGL(z1, dim3(16384), dim3(256), 0, 0, d, n4); });
  run("grid-stride 2048 blocks", [&] { hipLaunchKernelGGL(z1, dim3(2048), dim3(256), 0, 0, d, n4); });
  run("grid-stride 1024 blocks", [&] { hipLaunchKernelGGL(z1, dim3(1024), dim3(256), 0, 0, d, n4); });
  for (int nb : {16384, 4096, 2048, 1024}) {
    const int64_t pb = (n4 + nb - 1) / nb;
    char nm[64]; snprintf(nm, 64, "contiguous spans %d blocks", nb);
    run(nm, [&] { hipLaunchKernelGGL(z2, dim3(nb), dim3(256), 0, 0, d, n4, pb); });
    snprintf(nm, 64, "nontemporal spans %d", nb);
    run(nm, [&] { hipLaunchKernelGGL(z3, dim3(nb), dim3(256), 0, 0, d, n4, pb); });
  }
  run("hipMemsetAsync", [&] { hipMemsetAsync(d, 0, bytes, 0); });
  return 0;
}
