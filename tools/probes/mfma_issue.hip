// What does non-MFMA work cost a saturated fp32 MFMA loop on gfx950?  (kernel tuning aid)
//
// 2 waves / SIMD (256 registers each, like the 128x64 conv tile), every wave runs STEPS iterations of 512 MFMA cycles:
//   shape 0: 8  x v_mfma_f32_32x32x2_f32 (8 accumulators x 16 registers, 64 cycles each)
//   (64 KB of dynamic LDS per workgroup pins the occupancy at 2 workgroups per CU whatever the register count)
//   shape 1: 16 x v_mfma_f32_16x16x4_f32 (32 accumulators x 4 registers, 32 cycles each; two rounds over 16 of them)
// plus, per iteration, one of
//   extra 0: nothing
//   extra 1: 8 x v_mov_b32 into scratch registers (VALU register writes only)
//   extra 2: the conv k-step's memory instructions: 1 x buffer_load_dwordx4 + 2 x buffer_load_dword (cache-resident
//            lines, wave-uniform offset from the scalar unit) whose results are the MFMA operands two iterations later
//   extra 3: extra 2 + the two v_add_u32 of the address arithmetic
// The last lines repeat the 32x32x2 cases with 256 / 512 workgroups, i.e. 1 / 2 waves per SIMD in a single round.
// Prints TFLOP/s per combination:   hipcc --offload-arch=gfx950 -O3 -o mfma_issue mfma_issue.hip && ./mfma_issue
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int EXTRA>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ src, float* __restrict__ out, int steps, unsigned stride) {
    const int lane = threadIdx.x & 63;
    f32x16 acc32[8];
    f32x4 acc16[32];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc32[i][r] = 0.0f;
    for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) acc16[i][r] = 0.0f;
    f32x4 A[3];
    float B[3][2];
    const float* pa = src + lane * 4;
    const float* pb = src + 1024 + lane;
    unsigned voff = (unsigned)lane * 4u;
    for (int i = 0; i < 3; ++i) { A[i] = *(const f32x4*)(pa + 256 * i); B[i][0] = pb[64 * i]; B[i][1] = pb[64 * i + 512]; }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1 << 20, 0x00020000);
    float m0 = 1.0f, m1 = 2.0f, m2 = 3.0f, m3 = 4.0f, m4 = 5.0f, m5 = 6.0f, m6 = 7.0f, m7 = 8.0f;
    for (int s = 0; s < steps; s += 3) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const f32x4 a = A[u];
            const float b0 = B[u][0], b1 = B[u][1];
            if (SHAPE == 0) {
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb)
                        acc32[mb * 2 + nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb], nb ? b1 : b0, acc32[mb * 2 + nb], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    acc16[(u & 1) * 16 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i & 3], (i & 4) ? b1 : b0, acc16[(u & 1) * 16 + i], 0, 0, 0);
            }
            if (EXTRA == 1) {
                asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n"
                             "v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8"
                             : "=v"(m0), "=v"(m1), "=v"(m2), "=v"(m3), "=v"(m4), "=v"(m5), "=v"(m6), "=v"(m7) : "v"(b0));
            }
            if (EXTRA >= 2) {
                unsigned o0 = voff, o1 = voff;
                if (EXTRA == 3) {
                    asm volatile("v_add_u32 %0, %2, %3\n v_add_u32 %1, %2, %4" : "=v"(o0), "=v"(o1) : "s"(stride), "v"(voff), "v"(voff));
                }
                const int so = ((s + u) & 7) * 1024;            // scalar unit: the loads cannot be hoisted
                A[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(voff * 4u), so, 0));
                B[u][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)o0, so + 16384, 0));
                B[u][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)o1, so + 32768, 0));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float t = m0 + m1 + m2 + m3 + m4 + m5 + m6 + m7;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) t += acc32[i][r];
    for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) t += acc16[i][r];
    if (t == 12345.678f) out[threadIdx.x] = t;
}

template <int SHAPE, int EXTRA>
void run(const float* src, float* out, int wgs = 2048, int steps = 6000) {   // 2048 workgroups x 4 waves = 4 rounds of the 2048 resident waves
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k<SHAPE, EXTRA>), dim3(wgs), dim3(256), 65536, 0, src, out, steps, 0u);
    hipEventRecord(e0);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((k<SHAPE, EXTRA>), dim3(wgs), dim3(256), 65536, 0, src, out, steps, 0u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 3.0 * wgs * 4 * (double)steps * 8 * 4096;     // 512 MFMA cycles x 64 flops / cycle per wave-step
    printf("shape %s extra %d, %4d workgroups: %8.3f ms  %7.1f TFLOP/s\n", SHAPE ? "16x16x4" : "32x32x2", EXTRA, wgs, ms / 3, flops / (ms * 1e-3) / 1e12);
}

int main() {
    float *src, *out;
    hipMalloc(&src, 1 << 20); hipMemset(src, 0, 1 << 20);
    hipMalloc(&out, 1 << 16);
    run<0, 0>(src, out); run<0, 1>(src, out); run<0, 2>(src, out); run<0, 3>(src, out);
    run<1, 0>(src, out); run<1, 1>(src, out); run<1, 2>(src, out); run<1, 3>(src, out);
    // one workgroup per CU = a LONE wave on every SIMD (what a wave's neighbour sees while it is in its epilogue)
    run<0, 0>(src, out, 256, 24000); run<0, 2>(src, out, 256, 24000); run<0, 3>(src, out, 256, 24000);
    run<0, 0>(src, out, 512, 12000); run<0, 2>(src, out, 512, 12000);
    return 0;
}
