// Store-pattern probe (dev tool): how fast can a CU write an MFMA 32x32 fp32 accumulator tile to a (rows, positions) tensor?
//  A: the gather-GEMM epilogue pattern: lane (j = l & 31, h = l >> 5) stores 16 dwords, row = (r&3) + 8(r>>2) + 4h, 128 B per row and half-wave
//  B: the same bytes as dwordx4: lane owns 4 consecutive positions of one row (what an LDS transpose would give)
//  C: pattern A with non-temporal stores
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int V>
__global__ __launch_bounds__(256) void wr(float* out, int rows_total, int64_t rs, int ntiles) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    // tile t: 32 rows (row tile rt = t % (rows_total/32)) x 128 positions (pt = t / (rows_total/32))
    const int rtiles = rows_total / 32, rt = t % rtiles, pt = t / rtiles;
    float* base = out + (int64_t)(rt * 32) * rs + (int64_t)pt * 128;
    if (V == 0 || V == 2) {
      float* p = base + wave * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        if (V == 2) __builtin_nontemporal_store((float)(r + t), p + (int64_t)row * rs);
        else p[(int64_t)row * rs] = (float)(r + t);
      }
    } else {
      // wave w writes rows 8w..8w+7: lane -> (row = 8w + lane / 8, 16 positions... ) 128 positions = 32 quads; 8 rows x 32 quads = 256 -> 4 stores per lane
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = i * 64 + lane, row = wave * 8 + idx / 32, qd = idx % 32;
        f32x4 v = {(float)t, 1.f, 2.f, 3.f};
        *reinterpret_cast<f32x4*>(base + (int64_t)row * rs + 4 * qd) = v;
      }
    }
  }
}
int main() {
  const int rows = 96; const int64_t P = 8388608;      // 96 x 8.4M fp32 = 3.2 GB
  float* d; hipMalloc(&d, (size_t)rows * P * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int ntiles = (rows / 32) * (int)(P / 128);
  for (int v = 0; v < 3; ++v) for (int grid : {2048, 8192, 65536}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(a);
      if (v == 0) hipLaunchKernelGGL(wr<0>, dim3(grid), dim3(256), 0, 0, d, rows, P, ntiles);
      if (v == 1) hipLaunchKernelGGL(wr<1>, dim3(grid), dim3(256), 0, 0, d, rows, P, ntiles);
      if (v == 2) hipLaunchKernelGGL(wr<2>, dim3(grid), dim3(256), 0, 0, d, rows, P, ntiles);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (rep) printf("variant %d grid %6d: %.3f ms  %.2f TB/s\n", v, grid, ms, rows * P * 4.0 / ms / 1e9);
    }
  }
  return 0;
}
