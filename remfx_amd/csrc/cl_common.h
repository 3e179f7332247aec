// Channels-last bf16 family (round 5): shared device helpers.
//
// Layout.  A "cl" tensor is [N][A][B][C] with the C channels of one position contiguous and stored as bf16 (C % 8 == 0, so a
// position's channels are whole 16-byte groups).  The frequency branch of Hybrid Demucs uses A = frequency rows, B = 256 frames;
// 1-D tensors use A = 1.  Every kernel of the family reads and writes 16-byte channel groups: an MFMA B fragment (8 consecutive
// k = channels of one position) is one such group, so operands reach LDS by `buffer_load ... lds` DMA with no per-element work and
// results leave through an LDS transpose as full lines.
#pragma once
#include "common.h"

typedef __bf16 cl_bf16x8 __attribute__((ext_vector_type(8)));
typedef short cl_s16x4 __attribute__((ext_vector_type(4)));
#define CL_OOB 0x80000000u          // > num_records of every descriptor we build: the load returns 0 (into LDS too; probed on gfx950)
#define CL_LDS(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ __amdgpu_buffer_rsrc_t cl_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
// 64 lanes x 16 bytes from global straight into LDS at lds_base + lane * 16 (lds_base wave-uniform); out-of-range lanes land zeros
__device__ __forceinline__ void cl_glds16(__amdgpu_buffer_rsrc_t rs, unsigned char* lds_base, uint32_t voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, CL_LDS(lds_base), 16, voff, 0, 0, 0);
}
// The same DMA issued by inline asm, i.e. NOT tracked by the compiler.  With the builtin in flight hipcc puts `s_waitcnt vmcnt(0)` in
// front of the next LDS read it cannot prove disjoint from the DMA's target (found in the ISA of cl_wgrad's step loop and of the fused
// DConv kernels, round 6): a ring that is supposed to run several steps ahead is then drained at every step.  The issuing wave orders
// its own reads behind the pieces with counted `s_waitcnt vmcnt(n)` (CL_VMCNT; VMEM operations retire in order, and a compiler-
// generated counted wait can only become stricter through operations it does not know of) + a barrier for the other waves'.
// m0 is clobbered on purpose: a kernel that uses this must not also use the builtin (clang does not preserve reserved registers
// around asm, hence the silenced note).
typedef int cl_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ cl_i32x4 cl_rsrc_words(const void* p, uint32_t bytes) {
  const uint64_t a = (uint64_t)(uintptr_t)p;
  cl_i32x4 r = {(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xffffu), (int)bytes, 0x00020000};
  return r;
}
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void cl_glds16_quiet(const cl_i32x4& rs, unsigned char* lds_base, uint32_t voff) {
  const uint32_t m = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)CL_LDS(lds_base));
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(m), "v"(voff), "s"(rs) : "memory", "m0");
}
#pragma clang diagnostic pop
__device__ __forceinline__ float cl_bf2f(uint32_t bits16) { return __uint_as_float(bits16 << 16); }
// 8 bf16 (one 16-byte group) <-> 8 floats
__device__ __forceinline__ void cl_unpack8(const uint4& u, float (&v)[8]) {
  v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
  v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
  v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
  v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 cl_pack8(const float (&v)[8]) {
  return make_uint4(rfx_cvt_pk_bf16(v[0], v[1]), rfx_cvt_pk_bf16(v[2], v[3]), rfx_cvt_pk_bf16(v[4], v[5]),
                    rfx_cvt_pk_bf16(v[6], v[7]));
}
// wait for all but the newest n VMEM operations of this wave (LDS-DMA pieces included); n is a literal
#define CL_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define CL_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
