// How fast does a saturated v_mfma_f32_32x32x2_f32 stream run as a function of the number of accumulator chains it rotates over and
// of the operand DATA (zeros draw less power than random values)?  Round 6: three differently built kernels for 32 -> 32 channel
// layers (one 32-row band per wave = few accumulator tiles) all stop at ~106 TFLOP/s = 0.67 of the fp32 matrix peak.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_chain mfma_chain.hip && ./mfma_chain
// 2 waves per SIMD (64 KB of LDS per workgroup of 4 waves pins that), no memory instructions inside the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ src, float* __restrict__ out, int steps) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    float A[9], B[8];
    for (int i = 0; i < 9; ++i) A[i] = src[lane + 64 * i];
    for (int i = 0; i < 8; ++i) B[i] = src[1024 + lane + 64 * i];
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int u = 0; u < 72; ++u) {            // 72 MFMAs per iteration, accumulators in rotation
            acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[u % 9], B[u % 8], acc[u % NACC], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float t = 0.0f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) t += acc[i][r];
    if (t == 12345.678f) out[threadIdx.x] = t;
}

template <int NACC>
void run(const float* src, float* out, const char* what, int wgs = 2048, int steps = 700) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k<NACC>), dim3(wgs), dim3(256), 65536, 0, src, out, steps);
    hipEventRecord(e0);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((k<NACC>), dim3(wgs), dim3(256), 65536, 0, src, out, steps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 3.0 * wgs * 4 * (double)steps * 72 * 4096;
    printf("%d accumulator chains, %s, %4d workgroups: %8.3f ms  %7.1f TFLOP/s\n", NACC, what, wgs, ms / 3, flops / (ms * 1e-3) / 1e12);
}

int main() {
    float *src, *out;
    hipMalloc(&src, 1 << 20);
    hipMalloc(&out, 1 << 16);
    float* h = (float*)malloc(1 << 20);
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = 0; i < (1 << 18); ++i) h[i] = pass ? (float)rand() / RAND_MAX - 0.5f : 0.0f;
        hipMemcpy(src, h, 1 << 20, hipMemcpyHostToDevice);
        const char* what = pass ? "random operands" : "zero operands  ";
        run<1>(src, out, what); run<2>(src, out, what); run<4>(src, out, what); run<8>(src, out, what);
        run<2>(src, out, what, 512, 2800);       // one round of 2 waves per SIMD
        run<2>(src, out, what, 256, 5600);       // a lone wave per SIMD
    }
    return 0;
}
