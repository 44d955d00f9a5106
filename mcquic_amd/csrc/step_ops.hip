// Per-step bookkeeping of the training quantizer as a handful of launches (gfx950).
//
// The reference spells these out as strings of small tensor ops (mcquic/modules/quantizer.py:194-200: the code-usage
// exponent of `_randomDrop`; mcquic/modules/entropyCoder.py:28-44: one-hot counts, normalisation and the frequency EMA;
// mcquic/nn/base.py:17-29: LowerBound's gradient rule on the temperature).  Each of those is a 4-5 us launch of a few
// hundred elements; inside a captured training step there were 98 of them per replay.  Here:
//   vq_step_prologue_kernel    ONE launch in front of the level cascade: every level's drop exponent, every level's
//                              generator snapshot (+ the generator's advance), and the zeroing of the code-count buffer
//                              the sampling kernels add into (csrc/vq_train.hip)
//   vq_temperature_grad_kernel d temperature [m] from the per-row terms of the soft-max backward, LowerBound's rule inside
//   freq_ema_update_kernel     ONE launch for all levels: counts -> normalised -> EMA, in place on the `_freqEMA` parameters
//   nonneg_reparam_bwd2_kernel both re-parametrisation gradients of a GDN layer (beta [C], gamma [C, C]) in one launch
// All reductions run in a fixed order (thread-strided partial sums, then a tree over the workgroup): bit-reproducible.
#include "mcq_common.h"
#include "../../include/mcquic_hip.h"

namespace {

constexpr int STEP_T = 1024;                   // (a row of 8192 entries is 8 loads per thread, issued together)

struct StepPrologueK {
    const float* freq[MCQ_VQ_MAX_LEVELS];     // level l: [m_l, k_l] frequency EMA
    int32_t mk[MCQ_VQ_MAX_LEVELS];            // m_l * k_l
    float neg_bits_m1[MCQ_VQ_MAX_LEVELS];     // float(-(log2 k_l - 1))
    float bits[MCQ_VQ_MAX_LEVELS];            // float(log2 k_l)
    int32_t levels;
    float eps;
    float* exponents;                          // [levels]
    unsigned long long* rng_state;             // {seed, offset} or null
    unsigned long long* rng_snaps;             // [levels][2] or null
    unsigned long long* counts;                // zeroed, or null
    long long counts_n;
};

__device__ __forceinline__ int block_sum_int(int v, int* sm) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    int t = 0;
#pragma unroll
    for (int w = 0; w < STEP_T / 64; ++w) t += sm[w];
    __syncthreads();
    return t;
}

__device__ __forceinline__ float block_sum_float(float v, float* sm) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = sm[0];
#pragma unroll
    for (int w = 1; w < STEP_T / 64; ++w) t += sm[w];
    __syncthreads();
    return t;
}

// blocks [0, levels): one level's exponent each (block 0 also hands out the generator snapshots); the rest zero the counts
__global__ __launch_bounds__(STEP_T) void vq_step_prologue_kernel(StepPrologueK q) {
    __shared__ int sm[STEP_T / 64];
    const int b = blockIdx.x;
    if (b < q.levels) {
        const float* f = q.freq[b];
        const int n = q.mk[b];
        int cnt = 0;
        for (int i0 = 0; i0 < n; i0 += STEP_T * 4) {                  // four independent loads in flight per thread
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = i0 + e * STEP_T + (int)threadIdx.x;
                v[e] = i < n ? f[i] : 0.0f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) cnt += v[e] > q.eps ? 1 : 0;   // (eps > 0: the padding never counts)
        }
        cnt = block_sum_int(cnt, sm);
        if (threadIdx.x == 0) {
            // codeUsage = (freqEMA > eps).float().mean().clamp(0, 1): a count of ones divided by the element count
            float usage = (float)cnt / (float)n;
            usage = fminf(fmaxf(usage, 0.0f), 1.0f);
            // -(bits - 1) * codeUsage ** 2 + bits: torch rounds the square, the product and the sum one by one (-ffp-contract=off)
            const float sq = usage * usage;
            const float prod = q.neg_bits_m1[b] * sq;
            q.exponents[b] = prod + q.bits[b];
            if (b == 0 && q.rng_state && q.rng_snaps) {
                const unsigned long long seed = q.rng_state[0], off = q.rng_state[1];
                for (int l = 0; l < q.levels; ++l) {
                    q.rng_snaps[2 * l] = seed;
                    q.rng_snaps[2 * l + 1] = off + (unsigned long long)l;
                }
                q.rng_state[1] = off + (unsigned long long)q.levels;
            }
        }
        return;
    }
    if (q.counts) {
        const long long i0 = ((long long)(b - q.levels) * STEP_T + threadIdx.x) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (i0 + e < q.counts_n) q.counts[i0 + e] = 0ull;
    }
}

// dT[g] = mask * sum_{n, p} dtrow[n, g, p], mask = (T[g] >= bound) | (sum < 0)   (LowerBound.backward, nn/base.py:24-29)
__global__ __launch_bounds__(STEP_T) void vq_temperature_grad_kernel(const float* __restrict__ dtrow, const float* __restrict__ temperature,
                                                                     float bound, float* __restrict__ dt, int N, int m, int hw) {
    __shared__ float sm[STEP_T / 64];
    const int g = blockIdx.x;
    float acc = 0.0f;
    const int per = N * hw;
    for (int i = threadIdx.x; i < per; i += STEP_T) {
        const int n = i / hw, p = i - n * hw;
        acc += dtrow[((size_t)n * m + g) * hw + p];
    }
    acc = block_sum_float(acc, sm);
    if (threadIdx.x == 0) {
        const bool pass = temperature[g] >= bound || acc < 0.0f;
        dt[g] = (pass ? 1.0f : 0.0f) * acc;
    }
}

struct FreqEmaK {
    float* freq[MCQ_VQ_MAX_LEVELS];           // level l: [m_l, k_l], updated in place
    int32_t m[MCQ_VQ_MAX_LEVELS];
    int32_t k[MCQ_VQ_MAX_LEVELS];
    long long offset[MCQ_VQ_MAX_LEVELS];      // level l's first entry in `counts`
    int32_t row0[MCQ_VQ_MAX_LEVELS + 1];      // first (level, group) row of level l in blockIdx order
    int32_t levels;
    float keep, fresh;                         // float(ema), float(1 - ema)
    const long long* counts;
};

// one workgroup per (level, group) row: total, then freq = (1 - ema) * count / total + ema * freq, rounded op by op like torch
__global__ __launch_bounds__(STEP_T) void freq_ema_update_kernel(FreqEmaK q) {
    __shared__ long long sml[STEP_T / 64];
    int lv = 0;
    while (lv + 1 < q.levels && (int)blockIdx.x >= q.row0[lv + 1]) ++lv;
    const int g = (int)blockIdx.x - q.row0[lv];
    const int k = q.k[lv];
    const long long* c = q.counts + q.offset[lv] + (long long)g * k;
    float* f = q.freq[lv] + (size_t)g * k;
    long long tot = 0;
    for (int i0 = 0; i0 < k; i0 += STEP_T * 4) {
        long long v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = i0 + e * STEP_T + (int)threadIdx.x;
            v[e] = i < k ? c[i] : 0ll;
        }
        tot += (v[0] + v[1]) + (v[2] + v[3]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) tot += __shfl_xor(tot, off);
    if ((threadIdx.x & 63) == 0) sml[threadIdx.x >> 6] = tot;
    __syncthreads();
    tot = 0;
#pragma unroll
    for (int w = 0; w < STEP_T / 64; ++w) tot += sml[w];
    const float total = (float)tot;               // (counts are exact integers: their float sum is this conversion below 2^24)
    for (int i0 = 0; i0 < k; i0 += STEP_T * 4) {
        long long cv[4];
        float fv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = i0 + e * STEP_T + (int)threadIdx.x;
            cv[e] = i < k ? c[i] : 0ll;
            fv[e] = i < k ? f[i] : 0.0f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = i0 + e * STEP_T + (int)threadIdx.x;
            const float normalized = (float)cv[e] / total;
            const float a = q.fresh * normalized;
            const float b = q.keep * fv[e];
            if (i < k) f[i] = a + b;
        }
    }
}

struct Reparam2K {
    const float* p[2];
    const float* g[2];
    float* out[2];
    float bound[2];
    long long n[2];
};

__global__ __launch_bounds__(STEP_T) void nonneg_reparam_bwd2_kernel(Reparam2K q) {
    const long long i = (long long)blockIdx.x * STEP_T + threadIdx.x;
    const int which = i < q.n[0] ? 0 : 1;
    const long long j = which ? i - q.n[0] : i;
    if (j >= q.n[which]) return;
    const float p = q.p[which][j], bound = q.bound[which];
    const float gfold = (2.0f * fmaxf(p, bound)) * q.g[which][j];
    q.out[which][j] = (p >= bound || gfold < 0.0f) ? gfold : 0.0f;
}

}  // namespace

extern "C" int32_t mcq_vq_max_levels(void) { return MCQ_VQ_MAX_LEVELS; }

extern "C" int mcq_vq_step_prologue_f32(const float* const* freq_ema, const int32_t* m, const int32_t* k, int32_t levels, float eps,
                                        float* exponents, uint64_t* rng_state, uint64_t* rng_snaps, int64_t* counts, int64_t counts_n,
                                        void* stream) {
    if (!freq_ema || !m || !k || !exponents || levels <= 0 || levels > MCQ_VQ_MAX_LEVELS) return MCQ_EINVAL;
    if ((rng_state == nullptr) != (rng_snaps == nullptr)) return MCQ_EINVAL;
    if ((counts == nullptr) != (counts_n == 0) || counts_n < 0) return MCQ_EINVAL;
    StepPrologueK q;
    for (int l = 0; l < levels; ++l) {
        if (!freq_ema[l] || m[l] <= 0 || k[l] <= 0 || (long long)m[l] * k[l] > 0x7fffffffLL) return MCQ_EINVAL;
        q.freq[l] = freq_ema[l];
        q.mk[l] = m[l] * k[l];
        const double bits = log2((double)k[l]);                  // math.log2(k) (quantizer.py:108); the products below are float32 ops
        q.neg_bits_m1[l] = (float)(-(bits - 1.0));
        q.bits[l] = (float)bits;
    }
    q.levels = levels; q.eps = eps; q.exponents = exponents;
    q.rng_state = reinterpret_cast<unsigned long long*>(rng_state);
    q.rng_snaps = reinterpret_cast<unsigned long long*>(rng_snaps);
    q.counts = reinterpret_cast<unsigned long long*>(counts);
    q.counts_n = counts_n;
    const unsigned zero_blocks = (unsigned)((counts_n + STEP_T * 4 - 1) / (STEP_T * 4));
    hipLaunchKernelGGL(vq_step_prologue_kernel, dim3((unsigned)levels + zero_blocks), dim3(STEP_T), 0, (hipStream_t)stream, q);
    return mcq_check_launch();
}

extern "C" int mcq_vq_temperature_grad_f32(const float* dtrow, const float* temperature, float bound, float* dtemperature, int32_t N,
                                           int32_t m, int32_t hw, void* stream) {
    if (!dtrow || !temperature || !dtemperature || N <= 0 || m <= 0 || hw <= 0 || (long long)N * hw > 0x7fffffffLL) return MCQ_EINVAL;
    hipLaunchKernelGGL(vq_temperature_grad_kernel, dim3((unsigned)m), dim3(STEP_T), 0, (hipStream_t)stream, dtrow, temperature, bound,
                       dtemperature, N, m, hw);
    return mcq_check_launch();
}

extern "C" int mcq_freq_ema_update_f32(float* const* freq_ema, const int32_t* m, const int32_t* k, int32_t levels, const int64_t* counts,
                                       double ema, void* stream) {
    if (!freq_ema || !m || !k || !counts || levels <= 0 || levels > MCQ_VQ_MAX_LEVELS) return MCQ_EINVAL;
    FreqEmaK q;
    long long off = 0;
    int rows = 0;
    for (int l = 0; l < levels; ++l) {
        if (!freq_ema[l] || m[l] <= 0 || k[l] <= 0) return MCQ_EINVAL;
        q.freq[l] = freq_ema[l]; q.m[l] = m[l]; q.k[l] = k[l]; q.offset[l] = off; q.row0[l] = rows;
        off += (long long)m[l] * k[l];
        rows += m[l];
    }
    q.row0[levels] = rows;
    q.levels = levels;
    // (1 - ema) and ema reach torch's kernels as Python floats cast to float32 (entropyCoder.py:42)
    q.keep = (float)ema; q.fresh = (float)(1.0 - ema);
    q.counts = reinterpret_cast<const long long*>(counts);
    hipLaunchKernelGGL(freq_ema_update_kernel, dim3((unsigned)rows), dim3(STEP_T), 0, (hipStream_t)stream, q);
    return mcq_check_launch();
}

extern "C" int mcq_nonneg_reparam_bwd2_f32(const float* p0, const float* dfolded0, float bound0, float* dp0, int64_t n0, const float* p1,
                                           const float* dfolded1, float bound1, float* dp1, int64_t n1, void* stream) {
    if (!p0 || !dfolded0 || !dp0 || !p1 || !dfolded1 || !dp1 || n0 <= 0 || n1 <= 0) return MCQ_EINVAL;
    Reparam2K q;
    q.p[0] = p0; q.g[0] = dfolded0; q.out[0] = dp0; q.bound[0] = bound0; q.n[0] = n0;
    q.p[1] = p1; q.g[1] = dfolded1; q.out[1] = dp1; q.bound[1] = bound1; q.n[1] = n1;
    const long long total = n0 + n1;
    hipLaunchKernelGGL(nonneg_reparam_bwd2_kernel, dim3((unsigned)((total + STEP_T - 1) / STEP_T)), dim3(STEP_T), 0, (hipStream_t)stream, q);
    return mcq_check_launch();
}
