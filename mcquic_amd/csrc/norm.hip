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
#include <cstdlib>
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

// Statistics that workgroups of ONE launch hand to each other (the fused kernels below) are read at device scope -- the eight XCDs'
// L2s are not coherent with each other for ordinary accesses inside a kernel, and several runs' statistics share a cache line.
// (The two-launch form reads them in the next kernel: ordinary loads.)
template <bool COH> __device__ __forceinline__ float gn_stat_load(const float* p) {
    if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
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


// ---- large runs (round 5): many workgroups per (image, group) ---------------------------------------------------------------------
// One workgroup per run is fine while a run is a few hundred floats (4x4 ... 8x8 maps); Neon puts
// GroupNorm(32, 32) on 512 x 512 maps -- a run is one 1 MB plane, 4 images are 128 workgroups on 256 CUs each walking its plane three
// times with 4-byte loads, and the backward sums ran one WAVE per plane: a captured Neon training step spent 57 of its 92 ms here.
// Chunked form: a plane is cut into chunks of GN_CHUNK floats, one workgroup each.
//   forward   gn_chunk_stats_kernel  per chunk (mean_c, M2_c) two-pass over the chunk's own values (registers)
//             gn_chunk_apply_kernel  every workgroup merges its run's chunk statistics in chunk order (Chan's pairwise update: no
//                                    E[x^2] - E[x]^2 cancellation, deterministic), then normalises its chunk
//   backward  gn_chunk_bwd_sums_kernel / gn_chunk_bwd_dx_kernel the same way for (sum dy, sum dy x)
constexpr int GN_CHUNK = 8192;          // floats per chunk: 32 floats (8 x 16 bytes) per thread
constexpr int GN_CHUNK_MIN_HW = 256;    // planes below this stay on the one-workgroup-per-run kernels (a chunk would be mostly padding)

struct GnChunkK {
    const float* x; const float* dy; const float* gamma; const float* beta;
    float* y; float* y_silu; float* mean_out; float* rstd_out; float* dx;
    const float* mean; const float* rstd;
    float* stats;            // forward: [planes][chunks][2] (mean_c, M2_c); backward: [planes][chunks][2] (sum dy, sum dy x)
    float* sum_dy; float* sum_dyx;      // backward: per-plane totals for the parameter kernel
    int C, HW, groups, chunks;
    float eps;
};

// the chunk's values, 8 float4 per thread (zero beyond the plane), and how many of them are real
__device__ __forceinline__ int gn_load_chunk(const float* __restrict__ p, int HW, int chunk, f32x4v (&v)[GN_CHUNK / (4 * kThreads)]) {
    const int first = chunk * GN_CHUNK;
    const int n = HW - first < GN_CHUNK ? HW - first : GN_CHUNK;
    const bool vec = (((uintptr_t)(p + first) & 15) == 0);
#pragma unroll
    for (int e = 0; e < GN_CHUNK / (4 * kThreads); ++e) {
        const int i = (e * kThreads + (int)threadIdx.x) * 4;
        if (vec && i + 3 < n) v[e] = *reinterpret_cast<const f32x4v*>(p + first + i);
        else {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[e][q] = i + q < n ? p[first + i + q] : 0.0f;
        }
    }
    return n;
}

__global__ __launch_bounds__(kThreads) void gn_chunk_stats_kernel(GnChunkK k) {
    __shared__ float slots[4];
    const int plane = blockIdx.y, chunk = blockIdx.x;
    const float* xp = k.x + (size_t)plane * k.HW;
    f32x4v v[GN_CHUNK / (4 * kThreads)];
    const int n = gn_load_chunk(xp, k.HW, chunk, v);
    float s = 0.0f;
#pragma unroll
    for (int e = 0; e < GN_CHUNK / (4 * kThreads); ++e) s += (v[e][0] + v[e][1]) + (v[e][2] + v[e][3]);
    const float mean = block_sum(s, slots) / (float)n;
    float q = 0.0f;
#pragma unroll
    for (int e = 0; e < GN_CHUNK / (4 * kThreads); ++e) {
        const int i = (e * kThreads + (int)threadIdx.x) * 4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float a = i + t < n ? v[e][t] - mean : 0.0f;
            q += a * a;
        }
    }
    const float m2 = block_sum(q, slots);
    if (threadIdx.x == 0) {
        float* st = k.stats + ((size_t)plane * k.chunks + chunk) * 2;
        st[0] = mean; st[1] = m2;
    }
}

// (mean, rstd) of run `ng` from its chunk statistics, merged in (plane, chunk) order; every thread computes the same values
template <bool COH = false>
__device__ __forceinline__ void gn_merge_stats(const GnChunkK& k, int ng, float& mean, float& rstd) {
    const int cg = k.C / k.groups;
    const int n_img = ng / k.groups, g = ng % k.groups;
    const float* st = k.stats + ((size_t)(n_img * k.C + g * cg) * k.chunks) * 2;
    float cnt = 0.0f, mu = 0.0f, m2 = 0.0f;
    for (int pc = 0; pc < cg * k.chunks; ++pc) {
        const int chunk = pc % k.chunks;
        const int first = chunk * GN_CHUNK;
        const float nb = (float)(k.HW - first < GN_CHUNK ? k.HW - first : GN_CHUNK);
        const float mb = gn_stat_load<COH>(st + 2 * pc), qb = gn_stat_load<COH>(st + 2 * pc + 1);
        const float tot = cnt + nb, delta = mb - mu;
        mu = mu + delta * (nb / tot);
        m2 = m2 + qb + delta * delta * (cnt * nb / tot);
        cnt = tot;
    }
    mean = mu;
    rstd = 1.0f / sqrtf(m2 / cnt + k.eps);
}

__global__ __launch_bounds__(kThreads) void gn_chunk_apply_kernel(GnChunkK k) {
    const int plane = blockIdx.y, chunk = blockIdx.x;
    const int n_img = plane / k.C, ch = plane % k.C;
    const int cg = k.C / k.groups;
    const int ng = n_img * k.groups + ch / cg;
    float mean, rstd;
    gn_merge_stats(k, ng, mean, rstd);
    if (threadIdx.x == 0 && chunk == 0 && ch % cg == 0 && k.mean_out) { k.mean_out[ng] = mean; k.rstd_out[ng] = rstd; }
    const float scale = rstd * (k.gamma ? k.gamma[ch] : 1.0f);
    const float shift = __builtin_fmaf(-scale, mean, k.beta ? k.beta[ch] : 0.0f);
    const float* xp = k.x + (size_t)plane * k.HW;
    f32x4v v[GN_CHUNK / (4 * kThreads)];
    const int n = gn_load_chunk(xp, k.HW, chunk, v);
    const int first = chunk * GN_CHUNK;
    float* yp = k.y + (size_t)plane * k.HW + first;
    float* sp = k.y_silu ? k.y_silu + (size_t)plane * k.HW + first : nullptr;
    const bool vec = (((uintptr_t)yp & 15) == 0);
#pragma unroll
    for (int e = 0; e < GN_CHUNK / (4 * kThreads); ++e) {
        const int i = (e * kThreads + (int)threadIdx.x) * 4;
        f32x4v o, so;
#pragma unroll
        for (int t = 0; t < 4; ++t) { o[t] = __builtin_fmaf(v[e][t], scale, shift); so[t] = sp ? mcq_silu(o[t]) : 0.0f; }
        if (vec && i + 3 < n) {
            *reinterpret_cast<f32x4v*>(yp + i) = o;
            if (sp) *reinterpret_cast<f32x4v*>(sp + i) = so;
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (i + t < n) { yp[i + t] = o[t]; if (sp) sp[i + t] = so[t]; }
        }
    }
}

__global__ __launch_bounds__(kThreads) void gn_chunk_bwd_sums_kernel(GnChunkK k) {
    __shared__ float slots[8];
    const int plane = blockIdx.y, chunk = blockIdx.x;
    f32x4v xv[GN_CHUNK / (4 * kThreads)], dv[GN_CHUNK / (4 * kThreads)];
    gn_load_chunk(k.x + (size_t)plane * k.HW, k.HW, chunk, xv);
    gn_load_chunk(k.dy + (size_t)plane * k.HW, k.HW, chunk, dv);          // (zero beyond the plane: those terms add nothing)
    float a = 0.0f, b = 0.0f;
#pragma unroll
    for (int e = 0; e < GN_CHUNK / (4 * kThreads); ++e)
#pragma unroll
        for (int t = 0; t < 4; ++t) { a += dv[e][t]; b += dv[e][t] * xv[e][t]; }
    a = block_sum(a, slots);
    b = block_sum(b, slots + 4);
    if (threadIdx.x == 0) {
        float* st = k.stats + ((size_t)plane * k.chunks + chunk) * 2;
        st[0] = a; st[1] = b;
    }
}

__global__ __launch_bounds__(kThreads) void gn_chunk_bwd_dx_kernel(GnChunkK k) {
    const int plane = blockIdx.y, chunk = blockIdx.x;
    const int n_img = plane / k.C, ch = plane % k.C;
    const int cg = k.C / k.groups;
    const int g = ch / cg, ng = n_img * k.groups + g;
    // the run's sums: chunks in order per plane, planes in order (every thread, same order)
    float ds = 0.0f, db = 0.0f, own_dy = 0.0f, own_dyx = 0.0f;
    for (int c = 0; c < cg; ++c) {
        const int pl = n_img * k.C + g * cg + c;
        const float* st = k.stats + (size_t)pl * k.chunks * 2;
        float a = 0.0f, b = 0.0f;
        for (int q = 0; q < k.chunks; ++q) { a += st[2 * q]; b += st[2 * q + 1]; }
        const float gm = k.gamma ? k.gamma[g * cg + c] : 1.0f;
        ds += gm * b;
        db += gm * a;
        if (pl == plane) { own_dy = a; own_dyx = b; }
    }
    if (threadIdx.x == 0 && chunk == 0) { k.sum_dy[plane] = own_dy; k.sum_dyx[plane] = own_dyx; }
    const float mu = k.mean[ng], rs = k.rstd[ng];
    const float inv = 1.0f / (float)((long long)cg * k.HW);
    const float c2 = (db * mu - ds) * rs * rs * rs * inv;
    const float c3 = -c2 * mu - db * rs * inv;
    const float c1 = rs * (k.gamma ? k.gamma[ch] : 1.0f);
    f32x4v xv[GN_CHUNK / (4 * kThreads)], dv[GN_CHUNK / (4 * kThreads)];
    const int n = gn_load_chunk(k.x + (size_t)plane * k.HW, k.HW, chunk, xv);
    gn_load_chunk(k.dy + (size_t)plane * k.HW, k.HW, chunk, dv);
    const int first = chunk * GN_CHUNK;
    float* op = k.dx + (size_t)plane * k.HW + first;
    const bool vec = (((uintptr_t)op & 15) == 0);
#pragma unroll
    for (int e = 0; e < GN_CHUNK / (4 * kThreads); ++e) {
        const int i = (e * kThreads + (int)threadIdx.x) * 4;
        f32x4v o;
#pragma unroll
        for (int t = 0; t < 4; ++t) o[t] = c1 * dv[e][t] + c2 * xv[e][t] + c3;
        if (vec && i + 3 < n) *reinterpret_cast<f32x4v*>(op + i) = o;
        else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (i + t < n) op[i + t] = o[t];
        }
    }
}

// ---- one launch per direction (round 6) ---------------------------------------------------------------------------------------------
// The two launches above read every value twice: once for the chunk statistics, once to normalise.  A workgroup already holds its
// chunk in registers after the first read, so the fused kernels keep it there across a run-wide rendezvous: publish the chunk's pair of
// statistics as ONE 8-byte device-scope store, wait until the pairs of all cg * chunks workgroups of the run are there (a launch in
// front fills the table with a value no statistic can take; thread i polls entry i), merge in the same order as before and finish
// from the registers.  Same arithmetic, same order: bit-identical to the two-launch form.  No fences: a device-scope release /
// acquire pair writes back and invalidates the whole L2 of an XCD -- with a counter behind such fences the fused forward was 1.6x
// SLOWER than two launches (Neon dense inference 27.4 -> 42.6 ms per batch); self-validating entries need none.
// Measured with them (profiles/r06_group_norm_one_launch.txt): Neon(32, 4096, [16, 8, 4, 2, 2], denseNorm) encode + decode of 8 x 512x512
// 28.59 / 28.56 -> 28.18 / 30.94 ms, captured training step 39.82 -> 39.58 / 39.71 ms -- the second read of the two-launch form
// comes out of L2 / MALL and costs what the wait costs.  Bit-identical, tested (tests/test_gpu_step_ops.py), left OFF.
// Why waiting inside a kernel is safe here: workgroups are dispatched in linear order (chunks fastest, planes next), on every XCD in
// order, so the oldest unfinished run always has all its workgroups resident or next in line -- as long as a run is far smaller than
// what the chip holds (GN_FUSED_MAX_RUN workgroups against >= 2048 resident).  A wait that does not end within seconds traps
// instead of hanging the device.
constexpr int GN_FUSED_MAX_RUN = kThreads;
constexpr unsigned GN_PENDING_FWD = 0xbf800000u;        // M2 = -1.0f: a sum of squares is >= 0 or NaN
constexpr unsigned GN_PENDING_BWD = 0xffc0deadu;        // sum dy = a NaN no arithmetic produces (payloads only propagate from inputs)

__global__ void gn_fill_pending_kernel(unsigned long long* table, long long entries, unsigned long long pending) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < entries; i += (long long)gridDim.x * blockDim.x) table[i] = pending;
}

__device__ __forceinline__ unsigned long long gn_pack(float lo, float hi) {
    return (unsigned long long)__float_as_uint(lo) | ((unsigned long long)__float_as_uint(hi) << 32);
}

// publish this workgroup's pair (thread 0), then wait for the run's `need` pairs starting at `first` (thread i polls entry i)
template <bool BWD>
__device__ __forceinline__ void gn_run_rendezvous(unsigned long long* table, size_t mine, float lo, float hi, size_t first, int need) {
    if (threadIdx.x == 0) __hip_atomic_store(table + mine, gn_pack(lo, hi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((int)threadIdx.x < need) {
        unsigned spins = 0;
        for (;;) {
            const unsigned long long v = __hip_atomic_load(table + first + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned tag = BWD ? (unsigned)v : (unsigned)(v >> 32);
            if (tag != (BWD ? GN_PENDING_BWD : GN_PENDING_FWD)) break;
            __builtin_amdgcn_s_sleep(4);
            if (++spins > (1u << 22)) __builtin_trap();         // (seconds: cannot happen with in-order dispatch; abort rather than hang)
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(kThreads) void gn_chunk_fused_fwd_kernel(GnChunkK k) {
    __shared__ float slots[4];
    const int plane = blockIdx.y, chunk = blockIdx.x;
    const int n_img = plane / k.C, ch = plane % k.C;
    const int cg = k.C / k.groups;
    const int ng = n_img * k.groups + ch / cg;
    const float* xp = k.x + (size_t)plane * k.HW;
    f32x4v v[GN_CHUNK / (4 * kThreads)];
    const int n = gn_load_chunk(xp, k.HW, chunk, v);
    float s = 0.0f;
#pragma unroll
    for (int e = 0; e < GN_CHUNK / (4 * kThreads); ++e) s += (v[e][0] + v[e][1]) + (v[e][2] + v[e][3]);
    const float cmean = block_sum(s, slots) / (float)n;
    float q = 0.0f;
#pragma unroll
    for (int e = 0; e < GN_CHUNK / (4 * kThreads); ++e) {
        const int i = (e * kThreads + (int)threadIdx.x) * 4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float a = i + t < n ? v[e][t] - cmean : 0.0f;
            q += a * a;
        }
    }
    const float m2 = block_sum(q, slots);
    gn_run_rendezvous<false>(reinterpret_cast<unsigned long long*>(k.stats), (size_t)plane * k.chunks + chunk, cmean, m2,
                             (size_t)(n_img * k.C + (ch / cg) * cg) * k.chunks, cg * k.chunks);
    float mean, rstd;
    gn_merge_stats<true>(k, ng, mean, rstd);
    if (threadIdx.x == 0 && chunk == 0 && ch % cg == 0 && k.mean_out) { k.mean_out[ng] = mean; k.rstd_out[ng] = rstd; }
    const float scale = rstd * (k.gamma ? k.gamma[ch] : 1.0f);
    const float shift = __builtin_fmaf(-scale, mean, k.beta ? k.beta[ch] : 0.0f);
    const int first = chunk * GN_CHUNK;
    float* yp = k.y + (size_t)plane * k.HW + first;
    float* sp = k.y_silu ? k.y_silu + (size_t)plane * k.HW + first : nullptr;
    const bool vec = (((uintptr_t)yp & 15) == 0);
#pragma unroll
    for (int e = 0; e < GN_CHUNK / (4 * kThreads); ++e) {
        const int i = (e * kThreads + (int)threadIdx.x) * 4;
        f32x4v o, so;
#pragma unroll
        for (int t = 0; t < 4; ++t) { o[t] = __builtin_fmaf(v[e][t], scale, shift); so[t] = sp ? mcq_silu(o[t]) : 0.0f; }
        if (vec && i + 3 < n) {
            *reinterpret_cast<f32x4v*>(yp + i) = o;
            if (sp) *reinterpret_cast<f32x4v*>(sp + i) = so;
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (i + t < n) { yp[i + t] = o[t]; if (sp) sp[i + t] = so[t]; }
        }
    }
}

__global__ __launch_bounds__(kThreads) void gn_chunk_fused_bwd_kernel(GnChunkK k) {
    __shared__ float slots[8];
    const int plane = blockIdx.y, chunk = blockIdx.x;
    const int n_img = plane / k.C, ch = plane % k.C;
    const int cg = k.C / k.groups;
    const int g = ch / cg, ng = n_img * k.groups + g;
    f32x4v xv[GN_CHUNK / (4 * kThreads)], dv[GN_CHUNK / (4 * kThreads)];
    const int n = gn_load_chunk(k.x + (size_t)plane * k.HW, k.HW, chunk, xv);
    gn_load_chunk(k.dy + (size_t)plane * k.HW, k.HW, chunk, dv);          // (zero beyond the plane: those terms add nothing)
    float a0 = 0.0f, b0 = 0.0f;
#pragma unroll
    for (int e = 0; e < GN_CHUNK / (4 * kThreads); ++e)
#pragma unroll
        for (int t = 0; t < 4; ++t) { a0 += dv[e][t]; b0 += dv[e][t] * xv[e][t]; }
    a0 = block_sum(a0, slots);
    b0 = block_sum(b0, slots + 4);
    gn_run_rendezvous<true>(reinterpret_cast<unsigned long long*>(k.stats), (size_t)plane * k.chunks + chunk, a0, b0,
                            (size_t)(n_img * k.C + g * cg) * k.chunks, cg * k.chunks);
    float ds = 0.0f, db = 0.0f, own_dy = 0.0f, own_dyx = 0.0f;
    for (int c = 0; c < cg; ++c) {
        const int pl = n_img * k.C + g * cg + c;
        const float* st = k.stats + (size_t)pl * k.chunks * 2;
        float a = 0.0f, b = 0.0f;
        for (int q = 0; q < k.chunks; ++q) { a += gn_stat_load<true>(st + 2 * q); b += gn_stat_load<true>(st + 2 * q + 1); }
        const float gm = k.gamma ? k.gamma[g * cg + c] : 1.0f;
        ds += gm * b;
        db += gm * a;
        if (pl == plane) { own_dy = a; own_dyx = b; }
    }
    if (threadIdx.x == 0 && chunk == 0) { k.sum_dy[plane] = own_dy; k.sum_dyx[plane] = own_dyx; }
    const float mu = k.mean[ng], rs = k.rstd[ng];
    const float inv = 1.0f / (float)((long long)cg * k.HW);
    const float c2 = (db * mu - ds) * rs * rs * rs * inv;
    const float c3 = -c2 * mu - db * rs * inv;
    const float c1 = rs * (k.gamma ? k.gamma[ch] : 1.0f);
    const int first = chunk * GN_CHUNK;
    float* op = k.dx + (size_t)plane * k.HW + first;
    const bool vec = (((uintptr_t)op & 15) == 0);
#pragma unroll
    for (int e = 0; e < GN_CHUNK / (4 * kThreads); ++e) {
        const int i = (e * kThreads + (int)threadIdx.x) * 4;
        f32x4v o;
#pragma unroll
        for (int t = 0; t < 4; ++t) o[t] = c1 * dv[e][t] + c2 * xv[e][t] + c3;
        if (vec && i + 3 < n) *reinterpret_cast<f32x4v*>(op + i) = o;
        else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (i + t < n) op[i + t] = o[t];
        }
    }
}

inline int gn_chunks(int HW) { return (HW + GN_CHUNK - 1) / GN_CHUNK; }
// (every plane of >= 256 pixels: even where a plane is one partial chunk -- 64 x 64 maps -- a workgroup that reads its values once
//  with 16-byte loads and keeps them in registers beats the one-workgroup kernels' three scalar passes: 128 x 128 planes 245 -> ~20 us)
inline bool gn_chunked(int C, int HW, int groups) { (void)C; (void)groups; return HW >= GN_CHUNK_MIN_HW; }
inline unsigned long long gn_pack_host(unsigned lo, unsigned hi) { return (unsigned long long)lo | ((unsigned long long)hi << 32); }
// the one-launch form, OFF unless MCQUIC_AMD_GN_FUSED=1 is in the environment (read once per process): runs of at most GN_FUSED_MAX_RUN workgroups
inline bool gn_fused(int C, int HW, int groups) {
    static const bool on = [] { const char* e = getenv("MCQUIC_AMD_GN_FUSED"); return e && e[0] == '1'; }();
    return on && gn_chunked(C, HW, groups) && (long long)(C / groups) * gn_chunks(HW) <= GN_FUSED_MAX_RUN;
}

}  // namespace

extern "C" size_t mcq_group_norm_workspace_floats(int32_t N, int32_t C, int32_t HW, int32_t groups) {
    if (N <= 0 || C <= 0 || HW <= 0 || groups <= 0 || C % groups != 0) return 0;
    return gn_chunked(C, HW, groups) ? (size_t)N * C * gn_chunks(HW) * 2 : 0;
}

extern "C" int mcq_group_norm_f32(const float* x, const float* gamma, const float* beta, float* y, float* y_silu, float* mean_out,
                                  float* rstd_out, float* workspace, int32_t N, int32_t C, int32_t HW, int32_t groups, float eps, void* stream) {
    if (!x || !y || N <= 0 || C <= 0 || HW <= 0 || groups <= 0 || C % groups != 0 || !(eps >= 0.0f)) return MCQ_EINVAL;
    if ((mean_out == nullptr) != (rstd_out == nullptr)) return MCQ_EINVAL;
    if ((long long)(C / groups) * HW > 0x7fffffffLL || (long long)N * groups > 0x7fffffffLL) return MCQ_ETOOLARGE;
    if (workspace && gn_chunked(C, HW, groups) && (long long)N * C <= 65535) {        // many workgroups per run (see above)
        GnChunkK k = {};
        k.x = x; k.gamma = gamma; k.beta = beta; k.y = y; k.y_silu = y_silu; k.mean_out = mean_out; k.rstd_out = rstd_out;
        k.stats = workspace; k.C = C; k.HW = HW; k.groups = groups; k.chunks = gn_chunks(HW); k.eps = eps;
        const dim3 grid((unsigned)k.chunks, (unsigned)(N * C));
        if (gn_fused(C, HW, groups)) {
            const long long entries = (long long)N * C * k.chunks;
            hipLaunchKernelGGL(gn_fill_pending_kernel, dim3((unsigned)((entries + kThreads - 1) / kThreads < 256 ? (entries + kThreads - 1) / kThreads : 256)),
                               dim3(kThreads), 0, (hipStream_t)stream, reinterpret_cast<unsigned long long*>(workspace), entries, gn_pack_host(0u, GN_PENDING_FWD));
            hipLaunchKernelGGL(gn_chunk_fused_fwd_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, k);
            return mcq_check_launch();
        }
        hipLaunchKernelGGL(gn_chunk_stats_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, k);
        hipLaunchKernelGGL(gn_chunk_apply_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, k);
        return mcq_check_launch();
    }
    hipLaunchKernelGGL(group_norm_fwd_kernel, dim3((unsigned)(N * groups)), dim3(kThreads), 0, (hipStream_t)stream, x, gamma, beta, y,
                       y_silu, mean_out, rstd_out, C, HW, groups, eps);
    return mcq_check_launch();
}

extern "C" size_t mcq_group_norm_bwd_workspace_floats(int32_t N, int32_t C, int32_t HW, int32_t groups) {
    if (N <= 0 || C <= 0 || HW <= 0 || groups <= 0 || C % groups != 0) return 0;
    return (size_t)2 * N * C + (gn_chunked(C, HW, groups) ? (size_t)N * C * gn_chunks(HW) * 2 : 0);
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
    if (gn_chunked(C, HW, groups) && (long long)N * C <= 65535) {
        GnChunkK k = {};
        k.x = x; k.dy = dy; k.gamma = gamma; k.mean = mean; k.rstd = rstd; k.dx = dx;
        k.stats = workspace + (size_t)2 * N * C; k.sum_dy = sum_dy; k.sum_dyx = sum_dyx;
        k.C = C; k.HW = HW; k.groups = groups; k.chunks = gn_chunks(HW);
        const dim3 grid((unsigned)k.chunks, (unsigned)planes);
        if (gn_fused(C, HW, groups)) {
            const long long entries = (long long)N * C * k.chunks;
            hipLaunchKernelGGL(gn_fill_pending_kernel, dim3((unsigned)((entries + kThreads - 1) / kThreads < 256 ? (entries + kThreads - 1) / kThreads : 256)),
                               dim3(kThreads), 0, (hipStream_t)stream, reinterpret_cast<unsigned long long*>(k.stats), entries, gn_pack_host(GN_PENDING_BWD, 0u));
            hipLaunchKernelGGL(gn_chunk_fused_bwd_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, k);
        } else {
            hipLaunchKernelGGL(gn_chunk_bwd_sums_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, k);
            hipLaunchKernelGGL(gn_chunk_bwd_dx_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, k);
        }
        if (dgamma || dbeta)
            hipLaunchKernelGGL(group_norm_bwd_params_kernel, dim3((unsigned)((C + 127) / 128)), dim3(128), 0, (hipStream_t)stream, mean, rstd,
                               sum_dy, sum_dyx, dgamma, dbeta, N, C, groups);
        return mcq_check_launch();
    }
    hipLaunchKernelGGL(group_norm_bwd_sums_kernel, dim3((unsigned)((planes + 3) / 4)), dim3(kThreads), 0, (hipStream_t)stream, x, dy,
                       sum_dy, sum_dyx, planes, HW);
    hipLaunchKernelGGL(group_norm_bwd_dx_kernel, dim3((unsigned)(N * groups)), dim3(kThreads), 0, (hipStream_t)stream, x, dy, gamma, mean,
                       rstd, sum_dy, sum_dyx, dx, C, HW, groups);
    if (dgamma || dbeta)
        hipLaunchKernelGGL(group_norm_bwd_params_kernel, dim3((unsigned)((C + 127) / 128)), dim3(128), 0, (hipStream_t)stream, mean, rstd,
                           sum_dy, sum_dyx, dgamma, dbeta, N, C, groups);
    return mcq_check_launch();
}
