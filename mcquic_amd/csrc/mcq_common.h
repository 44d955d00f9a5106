// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of libmcquic_hip.
//
// Design notes (see DESIGN.md for the full picture):
//   * All contractions run on the exact-f32 matrix instruction v_mfma_f32_32x32x2_f32
//     (bitwise an fmaf chain; 64 cycles per SIMD, 157 TFLOP/s chip peak).  Because this
//     instruction is 16x slower than the bf16 forms, one MFMA (64 cycles) consumes only two
//     operand registers: operand bandwidth is tiny, so both operands are streamed straight from
//     global memory / L2 into VGPRs with a software prefetch ring -- no LDS staging, no barriers,
//     every wave is an independent stream.
//   * MFMA orientation: D[row = output channel][col = pixel].  Lane l supplies
//     A[i = l & 31][k = l >> 5] (weights) and B[k = l >> 5][j = l & 31] (activations), and owns
//     D[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31] for r in [0, 16).  With pixels on
//     the lane axis every NCHW load and store is a run of 32 consecutive floats.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

// voffset marker that is out of range for every buffer descriptor we build (planes < 2 GiB):
// the hardware returns 0 for such lanes, which is exactly the conv's zero padding.
#define MCQ_OOB 0x80000000u

// SiLU / sigmoid spelled like ATen's CPU kernels (x / (1 + exp(-x)), 1 / (1 + exp(-x))) with
// correctly rounded division; expf is the ocml implementation (~1 ulp).
#ifndef MCQ_FAST_ACT
#define MCQ_FAST_ACT 0
#endif
#if MCQ_FAST_ACT
// hardware transcendentals: exp(-x) = exp2(-x * log2(e)) (v_exp_f32, ~1 ulp) and v_rcp_f32 (1 ulp) instead of the
// ocml expf + IEEE division (~40 VALU instructions per value).
__device__ __forceinline__ float mcq_silu(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
__device__ __forceinline__ float mcq_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
#else
__device__ __forceinline__ float mcq_silu(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ float mcq_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
#endif

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mcq_make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ float mcq_buffer_load(__amdgpu_buffer_rsrc_t r, uint32_t voff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, 0, 0));
}

// Uniform (SGPR) 64-bit pointer from a possibly lane-tainted one.
template <typename T>
__device__ __forceinline__ T* mcq_uniform_ptr(T* p) {
    uint64_t v = (uint64_t)p;
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (T*)(((uint64_t)hi << 32) | lo);
}

// D-fragment row owned by register r of lane-half hi (32x32 MFMA accumulator map).
__device__ __forceinline__ int mcq_drow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

static inline int mcq_check_launch() { return hipGetLastError() == hipSuccess ? 0 : -2; }
