// GroupNorm for `denseNorm=True` (gfx950): the reference's ResidualBlock puts nn.GroupNorm(groups, C) in place of its second
// activation when `denseNorm` is set (mcquic/nn/blocks.py:179-200; `Neon(..., denseNorm)`, compressor.py:181-226).
//
// HBM-bound element-wise / reduction work -- no MFMA.  In NCHW the (C / groups) channels of a group are adjacent planes, so
// the elements one (image, group) normalises over are ONE contiguous run of cg * HW floats: a workgroup owns a run, reads it
// with 16-byte loads where alignment allows, and keeps the two-pass form (mean first, then centred squares: no
// E[x^2] - E[x]^2 cancellation).  The run is re-read for the second pass and for the normalisation; at the sizes on the
// path (<= 1 MB per run) those re-reads come out of L2.  Reductions: per-thread partial -> wave shuffle -> 4 slots in LDS,
// fixed order, deterministic.
//
// Forward (ATen's CPU kernel order, aten/src/ATen/native/cpu/group_norm_kernel.cpp): scale = rstd * gamma[c],
// shift = beta[c] - scale * mean, y = x * scale + shift, rstd = 1 / sqrt(var + eps) with the biased variance.
// Backward (the same file's formulas): with ds = sum_c gamma[c] sum_p dy x and db = sum_c gamma[c] sum_p dy over the run,
//   c2 = (db * mean - ds) * rstd^3 / count,  c3 = -c2 * mean - db * rstd / count,
//   dx = rstd * gamma[c] * dy + c2 * x + c3;   dgamma[c] = sum_n (sum_p dy x - mean sum_p dy) * rstd;  dbeta[c] = sum_n sum_p dy.
#include "mcq_common.h"
#include "../../include/mcquic_hip.h"

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// sum over the workgroup, result in every thread; `slots` = 4 floats of LDS per concurrent reduction
__device__ __forceinline__ float block_sum(float v, float* slots) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6;
    __syncthreads();                                   // slots may still be read from the previous reduction
    if ((threadIdx.x & 63) == 0) slots[wave] = v;
    __syncthreads();
    return (slots[0] + slots[1]) + (slots[2] + slots[3]);
}

// (mean, rstd) of one contiguous run; every thread returns the same values
__device__ __forceinline__ void run_moments(const float* __restrict__ x, int count, float eps, float* slots, float& mean, float& rstd) {
    const bool vec = (((uintptr_t)x & 15) == 0) && (count % 4 == 0);
    float s = 0.0f;
    if (vec) {
        const f32x4v* x4 = (const f32x4v*)x;
        for (int i = threadIdx.x; i < count / 4; i += kThreads) { const f32x4v v = x4[i]; s += (v[0] + v[1]) + (v[2] + v[3]); }
    } else {
        for (int i = threadIdx.x; i < count; i += kThreads) s += x[i];
    }
    mean = block_sum(s, slots) / (float)count;
    float q = 0.0f;
    if (vec) {
        const f32x4v* x4 = (const f32x4v*)x;
        for (int i = threadIdx.x; i < count / 4; i += kThreads) {
            const f32x4v v = x4[i];
            const float a = v[0] - mean, b = v[1] - mean, c = v[2] - mean, d = v[3] - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
    } else {
        for (int i = threadIdx.x; i < count; i += kThreads) { const float a = x[i] - mean; q += a * a; }
    }
    const float var = block_sum(q, slots) / (float)count;
    rstd = 1.0f / sqrtf(var + eps);
}

__global__ __launch_bounds__(kThreads) void group_norm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, float* __restrict__ y,
                                                                  float* __restrict__ y_silu, float* __restrict__ mean_out,
                                                                  float* __restrict__ rstd_out, int C, int HW, int groups, float eps) {
    __shared__ float slots[4];
    const int ng = blockIdx.x;                         // n * groups + g
    const int g = ng % groups, n = ng / groups;
    const int cg = C / groups;
    const size_t base = ((size_t)n * C + (size_t)g * cg) * HW;
    const int count = cg * HW;
    float mean, rstd;
    run_moments(x + base, count, eps, slots, mean, rstd);
    if (threadIdx.x == 0 && mean_out) { mean_out[ng] = mean; rstd_out[ng] = rstd; }
    for (int c = 0; c < cg; ++c) {
        const int ch = g * cg + c;
        const float scale = rstd * (gamma ? gamma[ch] : 1.0f);
        const float shift = __builtin_fmaf(-scale, mean, beta ? beta[ch] : 0.0f);
        const float* xp = x + base + (size_t)c * HW;
        float* yp = y + base + (size_t)c * HW;
        float* sp = y_silu ? y_silu + base + (size_t)c * HW : nullptr;
        for (int i = threadIdx.x; i < HW; i += kThreads) {
            const float v = __builtin_fmaf(xp[i], scale, shift);
            yp[i] = v;
            if (sp) sp[i] = mcq_silu(v);
        }
    }
}

// per (n, c) plane: sum_p dy and sum_p dy * x  (one wave per plane)
__global__ __launch_bounds__(kThreads) void group_norm_bwd_sums_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                       float* __restrict__ sum_dy, float* __restrict__ sum_dyx,
                                                                       int planes, int HW) {
    const int plane = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (plane >= planes) return;
    const int lane = threadIdx.x & 63;
    const float* xp = x + (size_t)plane * HW;
    const float* dp = dy + (size_t)plane * HW;
    float a = 0.0f, b = 0.0f;
    for (int i = lane; i < HW; i += 64) { const float d = dp[i]; a += d; b += d * xp[i]; }
    a = wave_sum(a);
    b = wave_sum(b);
    if (lane == 0) { sum_dy[plane] = a; sum_dyx[plane] = b; }
}

__global__ __launch_bounds__(kThreads) void group_norm_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                     const float* __restrict__ rstd, const float* __restrict__ sum_dy,
                                                                     const float* __restrict__ sum_dyx, float* __restrict__ dx,
                                                                     int C, int HW, int groups) {
    const int ng = blockIdx.x;
    const int g = ng % groups, n = ng / groups;
    const int cg = C / groups;
    const size_t base = ((size_t)n * C + (size_t)g * cg) * HW;
    float ds = 0.0f, db = 0.0f;                         // every thread: the same cg-term sums, same order
    for (int c = 0; c < cg; ++c) {
        const int ch = g * cg + c;
        const float gm = gamma ? gamma[ch] : 1.0f;
        ds += gm * sum_dyx[(size_t)n * C + ch];
        db += gm * sum_dy[(size_t)n * C + ch];
    }
    const float mu = mean[ng], rs = rstd[ng];
    const float inv = 1.0f / (float)(cg * HW);
    const float c2 = (db * mu - ds) * rs * rs * rs * inv;
    const float c3 = -c2 * mu - db * rs * inv;
    for (int c = 0; c < cg; ++c) {
        const float c1 = rs * (gamma ? gamma[g * cg + c] : 1.0f);
        const float* xp = x + base + (size_t)c * HW;
        const float* dp = dy + base + (size_t)c * HW;
        float* op = dx + base + (size_t)c * HW;
        for (int i = threadIdx.x; i < HW; i += kThreads) op[i] = c1 * dp[i] + c2 * xp[i] + c3;
    }
}

// dgamma[c] = sum_n (sum_dyx - mean sum_dy) rstd, dbeta[c] = sum_n sum_dy   (one thread per channel, images in order)
__global__ void group_norm_bwd_params_kernel(const float* __restrict__ mean, const float* __restrict__ rstd,
                                             const float* __restrict__ sum_dy, const float* __restrict__ sum_dyx,
                                             float* __restrict__ dgamma, float* __restrict__ dbeta, int N, int C, int groups) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= C) return;
    const int g = ch / (C / groups);
    float a = 0.0f, b = 0.0f;
    for (int n = 0; n < N; ++n) {
        const size_t p = (size_t)n * C + ch;
        a += (sum_dyx[p] - mean[n * groups + g] * sum_dy[p]) * rstd[n * groups + g];
        b += sum_dy[p];
    }
    if (dgamma) dgamma[ch] = a;
    if (dbeta) dbeta[ch] = b;
}

}  // namespace

extern "C" int mcq_group_norm_f32(const float* x, const float* gamma, const float* beta, float* y, float* y_silu, float* mean_out,
                                  float* rstd_out, int32_t N, int32_t C, int32_t HW, int32_t groups, float eps, void* stream) {
    if (!x || !y || N <= 0 || C <= 0 || HW <= 0 || groups <= 0 || C % groups != 0 || !(eps >= 0.0f)) return MCQ_EINVAL;
    if ((mean_out == nullptr) != (rstd_out == nullptr)) return MCQ_EINVAL;
    if ((long long)(C / groups) * HW > 0x7fffffffLL || (long long)N * groups > 0x7fffffffLL) return MCQ_ETOOLARGE;
    hipLaunchKernelGGL(group_norm_fwd_kernel, dim3((unsigned)(N * groups)), dim3(kThreads), 0, (hipStream_t)stream, x, gamma, beta, y,
                       y_silu, mean_out, rstd_out, C, HW, groups, eps);
    return mcq_check_launch();
}

extern "C" size_t mcq_group_norm_bwd_workspace_floats(int32_t N, int32_t C) {
    return N > 0 && C > 0 ? (size_t)2 * N * C : 0;
}

extern "C" int mcq_group_norm_bwd_f32(const float* x, const float* dy, const float* gamma, const float* mean, const float* rstd,
                                      float* dx, float* dgamma, float* dbeta, float* workspace, int32_t N, int32_t C, int32_t HW,
                                      int32_t groups, void* stream) {
    if (!x || !dy || !mean || !rstd || !dx || !workspace || N <= 0 || C <= 0 || HW <= 0 || groups <= 0 || C % groups != 0)
        return MCQ_EINVAL;
    if ((long long)(C / groups) * HW > 0x7fffffffLL || (long long)N * C > 0x7fffffffLL) return MCQ_ETOOLARGE;
    float* sum_dy = workspace;
    float* sum_dyx = workspace + (size_t)N * C;
    const int planes = N * C;
    hipLaunchKernelGGL(group_norm_bwd_sums_kernel, dim3((unsigned)((planes + 3) / 4)), dim3(kThreads), 0, (hipStream_t)stream, x, dy,
                       sum_dy, sum_dyx, planes, HW);
    hipLaunchKernelGGL(group_norm_bwd_dx_kernel, dim3((unsigned)(N * groups)), dim3(kThreads), 0, (hipStream_t)stream, x, dy, gamma, mean,
                       rstd, sum_dy, sum_dyx, dx, C, HW, groups);
    if (dgamma || dbeta)
        hipLaunchKernelGGL(group_norm_bwd_params_kernel, dim3((unsigned)((C + 127) / 128)), dim3(128), 0, (hipStream_t)stream, mean, rstd,
                           sum_dy, sum_dyx, dgamma, dbeta, N, C, groups);
    return mcq_check_launch();
}
