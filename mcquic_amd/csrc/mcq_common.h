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

// SiLU / sigmoid: x * r and r with r = 1 / (1 + exp(-x)), evaluated on the hardware transcendental units with
// the rounding errors that matter compensated in software (about a dozen VALU instructions per value; the ocml
// expf + IEEE divide sequence they replace costs ~40 and made the conv epilogues 10 % of a launch):
//   * exp(-x) = 2^t, t = -x log2(e).  t is split as t_hi + t_lo (the product's rounding error, recovered with an
//     fma, plus the low word of log2 e) so that v_exp_f32 (1 ulp) sees an exactly representable argument and
//     the lost bits come back through the first-order factor 1 + t_lo ln 2;
//   * n / d: v_rcp_f32 (1 ulp), q = n r, then one residual correction q += (n - d q) r.
// Measured against float64 over [-30, 30] (tests/test_gpu_ops.py::test_silu_accuracy, tools/probe_silu_accuracy.py):
// mean error 0.36 ulp, 99.9 % within 1.7 ulp, worst 3.3 ulp (at x ~ -16.7 where |silu| ~ 1e-6); ATen's float32 CPU
// kernel (Sleef expf, an add and a divide) measures 0.33 / 1.4 / 2.4 on the same points.  -DMCQ_EXACT_ACT=1 restores
// the ocml spelling.
#ifndef MCQ_EXACT_ACT
#define MCQ_EXACT_ACT 0
#endif
#if MCQ_EXACT_ACT
__device__ __forceinline__ float mcq_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float mcq_silu(float x) { return x / (1.0f + expf(-x)); }
#else
// 1 + exp(-x), finite for every finite x
__device__ __forceinline__ float mcq_one_plus_exp_neg(float x) {
    const float c_hi = -1.44269502162933349609375f;        // -float(log2 e)
    const float c_lo = -1.925963033500971e-8f;              // -(log2 e - float(log2 e))
    float t = x * c_hi;
    const float tl = __builtin_fmaf(x, c_lo, __builtin_fmaf(x, c_hi, -t));
    t = __builtin_fminf(t, 126.0f);                          // 1 / (1 + 2^126) is already 0 in effect
    float e = __builtin_amdgcn_exp2f(t);
    e = __builtin_fmaf(e, tl * 0.693147182464599609375f, e);
    return 1.0f + e;
}
// n / d from v_rcp_f32 and one residual correction (the quotient lands within ~0.5 ulp)
__device__ __forceinline__ float mcq_div(float n, float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    const float q = n * r;
    return __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q);
}
__device__ __forceinline__ float mcq_sigmoid(float x) { return mcq_div(1.0f, mcq_one_plus_exp_neg(x)); }
__device__ __forceinline__ float mcq_silu(float x) { return mcq_div(x, mcq_one_plus_exp_neg(x)); }
#endif

// silu of two values at once: the same IEEE operations as mcq_silu, component by component, with the nine multiplies / FMAs /
// adds as packed instructions (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32) -- 15 issue slots per pair instead of 24.  In an
// epilogue every VALU instruction takes the matrix pipe away from the co-resident wave for a few cycles.
#if MCQ_EXACT_ACT
__device__ __forceinline__ f32x2v mcq_silu2(f32x2v x) { return f32x2v{mcq_silu(x[0]), mcq_silu(x[1])}; }
#else
__device__ __forceinline__ f32x2v mcq_silu2(f32x2v x) {
    const f32x2v c_hi = {-1.44269502162933349609375f, -1.44269502162933349609375f};
    const f32x2v c_lo = {-1.925963033500971e-8f, -1.925963033500971e-8f};
    const f32x2v ln2 = {0.693147182464599609375f, 0.693147182464599609375f};
    f32x2v t = x * c_hi;
    const f32x2v tl = __builtin_elementwise_fma(x, c_lo, __builtin_elementwise_fma(x, c_hi, -t));
    t = f32x2v{__builtin_fminf(t[0], 126.0f), __builtin_fminf(t[1], 126.0f)};
    f32x2v e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
    e = __builtin_elementwise_fma(e, tl * ln2, e);
    const f32x2v d = f32x2v{1.0f, 1.0f} + e;
    const f32x2v r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    const f32x2v q = x * r;
    return __builtin_elementwise_fma(__builtin_elementwise_fma(-d, q, x), r, q);
}
#endif

// silu'(x) = s (1 + x (1 - s)), s = sigmoid(x)
__device__ __forceinline__ float mcq_dsilu(float x) {
    const float s = mcq_sigmoid(x);
    return s * (1.0f + x * (1.0f - s));
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mcq_make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ float mcq_buffer_load(__amdgpu_buffer_rsrc_t r, uint32_t voff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, 0, 0));
}

// Buffer accesses with a wave-uniform byte offset in soffset next to the per-lane voffset (the hardware range-checks
// their sum against num_records: out-of-range loads return 0, out-of-range stores are dropped).
__device__ __forceinline__ float mcq_buffer_load_s(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void mcq_buffer_store_s(float v, __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, (int)voff, (int)soff, 0);
}
typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2v mcq_buffer_load2_s(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(f32x2v, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void mcq_buffer_store2_s(f32x2v v, __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, v), r, (int)voff, (int)soff, 0);
}

typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4v mcq_buffer_load4_s(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void mcq_buffer_store4_s(f32x4v v, __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), r, (int)voff, (int)soff, 0);
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
