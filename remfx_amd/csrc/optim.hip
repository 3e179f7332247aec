// Optimiser step of the training path on one flat parameter / gradient buffer:
// global gradient-norm clipping (Lightning gradient_clip_val=10.0, cfg/config.yaml:119)
// and AdamW (models.py:185-191: betas (0.95, 0.999), eps 1e-6, weight_decay 1e-3).
// HBM-bound: 4 streams read (p, g, m, v), 3 written, float4 grid-stride.
#include "common.h"
#include <cstdlib>

// rfx_zero: buffers up to this size are reduction targets (statistics, loss sums, counters) and are filled write-through
constexpr int64_t RFX_ZERO_WT_MAX_BYTES = 4 << 20;

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ out) {
  double acc = 0.0;
  const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
    acc += (double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2] + (double)v[3] * v[3];
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) acc += (double)g[i] * g[i];
  const double v[1] = {acc};
  rfx_block_store_slot<1>(v, out, 0, gridDim.x, blockIdx.x);        // out = per-workgroup slots (rfx_sumsq adds them in order)
}

// gscale_ptr (device, optional): every gradient is multiplied by *gscale_ptr first (clip coefficient
// and/or 1/world_size), so clipping needs no host round trip.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                    float lr, float b1, float b2, float eps, float wd,
                                                    float bc1, float bc2, const float* __restrict__ gscale_ptr) {
  const float gs = gscale_ptr ? *gscale_ptr : 1.f;
  const float step = lr / bc1, rbc2 = 1.f / sqrtf(bc2), decay = 1.f - lr * wd;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float gi = g[i] * gs;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] * decay - step * mi / (sqrtf(vi) * rbc2 + eps);
  }
}

// Zero fill of nbytes (a multiple of 4, base 4-byte aligned): 16-byte stores, grid sized to the buffer.  (The framework's own fill ran
// the 334 MB gradient buffer at 0.5 TB/s and the step issues ~200 fills: 3 ms of a 150 ms step.)
__global__ __launch_bounds__(256) void zero_kernel(uint32_t* __restrict__ p, int64_t nwords) {
  const int64_t head = min<int64_t>(((16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15) >> 2, nwords);
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
  if (gid < head) p[gid] = 0u;
  uint4* q = reinterpret_cast<uint4*>(p + head);
  const int64_t n4 = (nwords - head) >> 2;
  for (int64_t i = gid; i < n4; i += stride) q[i] = make_uint4(0u, 0u, 0u, 0u);
  const int64_t tail0 = head + (n4 << 2);
  if (tail0 + gid < nwords) p[tail0 + gid] = 0u;
}

// Write-through zero fill: every word leaves as an agent-scope store (`global_store_dword ... sc1`), i.e. it is performed at the
// memory side of the XCDs' L2s instead of sitting in one XCD's L2 as a dirty line until the end-of-kernel write-back.  For buffers
// that agent-scope ATOMICS of a following kernel accumulate into (those execute at memory too): DESIGN.md 4.10.
__global__ __launch_bounds__(256) void zero_wt_kernel(uint32_t* __restrict__ p, int64_t nwords) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += stride)
    __hip_atomic_store(p + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// coef = min(1, max_norm / (sqrt(sumsq * pre^2) + 1e-6)) * pre      (torch clip_grad_norm_ semantics;
// pre = scale already owed to the gradients, e.g. 1/world_size after a summing all-reduce)
__global__ void clip_coef_kernel(const double* __restrict__ sumsq, float max_norm, float pre, float* __restrict__ coef,
                                 float* __restrict__ norm_out) {
  const float norm = (float)sqrt(*sumsq) * pre;
  if (norm_out) *norm_out = norm;
  float c = max_norm > 0.f ? max_norm / (norm + 1e-6f) : 1.f;
  *coef = (c < 1.f ? c : 1.f) * pre;
}

static int gridn(int64_t n) {
  const int64_t b = (n + 1023) / 1024;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

extern "C" int rfx_zero(void* p, int64_t nbytes, void* stream) {
  if (nbytes < 0 || (nbytes & 3) || (nbytes && !p) || (reinterpret_cast<uintptr_t>(p) & 3)) return -1;
  if (nbytes == 0) return 0;
  const int64_t nwords = nbytes >> 2;
  static const int wt_mode = [] { const char* e = getenv("RFX_ZERO_WT"); return e ? atoi(e) : 0; }();   // dev A/B (scripts/probes/zero_wt_ab.sh): 0 never (default), 1 small buffers, 2 always
  if (wt_mode == 2 || (wt_mode == 1 && nbytes <= (int64_t)RFX_ZERO_WT_MAX_BYTES)) {
    const int64_t bw = (nwords + 1023) / 1024;
    hipLaunchKernelGGL(zero_wt_kernel, dim3((unsigned)(bw < 1 ? 1 : (bw > 4096 ? 4096 : bw))), dim3(256), 0, (hipStream_t)stream,
                       static_cast<uint32_t*>(p), nwords);
    RFX_CHECK_LAUNCH();
    return 0;
  }
  const int64_t b = (nwords + 4095) / 4096;                       // four 16-byte stores per thread
  hipLaunchKernelGGL(zero_kernel, dim3((unsigned)(b < 1 ? 1 : (b > 16384 ? 16384 : b))), dim3(256), 0, (hipStream_t)stream,
                     static_cast<uint32_t*>(p), nwords);
  RFX_CHECK_LAUNCH();
  return 0;
}
// out[0] = sum g^2 (fp64); ws: RFX_SUMSQ_SLOTS doubles of per-workgroup partials (no initialisation needed), added in slot order
extern "C" int rfx_sumsq(const float* g, int64_t n, double* ws, double* out, void* stream) {
  if (!g || !out || !ws || n < 0) return -1;
  const int gr = n > 0 ? gridn(n) : 0;                                 // <= RFX_SUMSQ_SLOTS
  if (gr) {
    hipLaunchKernelGGL(sumsq_kernel, dim3(gr), dim3(256), 0, (hipStream_t)stream, g, n, ws);
    RFX_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(rfx_slot_sum_kernel<double>, RFX_SLOT_SUM_GRID(1), 0, (hipStream_t)stream, ws, 1, gr, 1, out);
  RFX_CHECK_LAUNCH();
  return 0;
}
extern "C" int rfx_clip_coef(const double* sumsq, float max_norm, float pre, float* coef, float* norm_out,
                             void* stream) {
  if (!sumsq || !coef) return -1;
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sumsq, max_norm, pre, coef, norm_out);
  RFX_CHECK_LAUNCH();
  return 0;
}
extern "C" int rfx_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1,
                              float b2, float eps, float wd, int32_t step, const float* gscale, void* stream) {
  if (!p || !g || !m || !v || n < 0 || step < 1) return -1;
  if (n == 0) return 0;
  const float bc1 = 1.f - powf(b1, (float)step), bc2 = 1.f - powf(b2, (float)step);
  hipLaunchKernelGGL(adamw_kernel, dim3(gridn(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, b1, b2,
                     eps, wd, bc1, bc2, gscale);
  RFX_CHECK_LAUNCH();
  return 0;
}
