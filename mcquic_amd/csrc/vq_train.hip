// Training-mode quantizer forward for gfx950 (BASELINE config #5, forward half):
//
//   mcq_vq_logits_f32          logit[n, g, y, x, k] = (-dist / sqrt(k)) * max(T_g, bound)
//                              (reference: mcquic/modules/quantizer.py:181-183,204 `_logit` * LowerBound(temperature))
//   mcq_vq_gumbel_sample_f32   random drop (:194-200) + gumbelSoftmax(hard=True) (mcquic/nn/base.py:118-133) + argmax
//                              code (:232-239), one wave per latent vector, the uniform draws are inputs
//   mcq_vq_dequant_soft_f32    sample @ codebook (:262-274) for the straight-through sample, which is exactly
//                              one-hot in value: v_hot * codebook[g, index]
//
// The logits kernel is the assign kernel's GEMM with the MFMA operands swapped -- D[row = latent vector][col =
// codeword] -- so that a lane owns ONE codeword for 16 vectors and the [.., k]-contiguous logits are stored as
// runs of 32 consecutive floats.  |x|^2 is brought into the accumulator's ROW layout by the same exact one-MFMA
// broadcast the assign kernel uses for |c|^2; |c|^2 is per column = per lane here.
#include "mcq_common.h"
#include "vq_common.h"
#include "vq_bwd_mfma.h"
#include "../../include/mcquic_hip.h"
#include <math.h>

namespace {

#ifndef MCQ_LOGITS_WAVES
#define MCQ_LOGITS_WAVES 2048      // waves a logits launch aims for before it stops splitting the codeword tiles over grid.z
#endif

struct VqLogitK {
    VqK v;
    const float* temperature;   // [m]
    float bound;
    float scale;                // sqrt(k)
    float rscale;               // RN(1 / scale)
    float* logits;              // [N, m, h, w, k]
    int raw;                    // 1: store the inner products <x_v, c_k> themselves (backward: dSample = dDeq . C^T)
    int tiles_per_z;            // codeword tiles per blockIdx.z (the logits of different tiles are independent)
};

// a / s, correctly rounded, for a divisor that is the same for the whole launch (sqrt(k)): with r = RN(1 / s), q0 = RN(a r) is
// within one ulp of the quotient, the remainder a - q0 s is exact in an fma, and RN(q0 + rem r) is the correctly rounded quotient
// (Markstein's division theorem; denormal quotients aside, which a distance over sqrt(k) never is).  Three instructions for the
// ~13 of the IEEE division sequence -- 33.5 M of them sit in the epilogue of the k = 8192 level, issued in the MFMAs' shadow-less
// tail of every tile (tests/test_gpu_train_forward.py::test_logit_division_is_the_ieee_quotient holds it to numpy's float32 division).
__device__ __forceinline__ float div_by_constant(float a, float s, float r) {
    const float q0 = a * r;
    const float rem = __builtin_fmaf(-q0, s, a);
    const float q1 = __builtin_fmaf(rem, r, q0);
    return __builtin_isfinite(q0) ? q1 : q0;
}

__global__ __launch_bounds__(256) void vq_logits_kernel(VqLogitK q) {
    const VqK& p = q.v;
    constexpr int NP = VQ_NB, NW = VQ_MB, PF = VQ_PF;     // 2 blocks of 32 vectors x 4 blocks of 32 codewords
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int gw = blockIdx.x * 4 + wave;
    if (gw * NP >= p.total_blocks) return;
    const int g = blockIdx.y;
    const int hi = lane >> 5, j = lane & 31;
    const int BW = 1 << p.bw_log2;
    const int ly = j >> p.bw_log2, lx = j & (BW - 1);
    const int BH = 32 >> p.bw_log2;
    const int HW = p.h * p.w;
    const unsigned group_bytes = (unsigned)p.d * (unsigned)HW * 4u;

    int img[NP], by0[NP], bx0[NP];
    bool blk[NP], valid[NP];
    unsigned pixoff[NP];
    __amdgpu_buffer_rsrc_t rsrc[NP];
#pragma unroll
    for (int nb = 0; nb < NP; ++nb) {
        int pb = gw * NP + nb;
        blk[nb] = pb < p.total_blocks;
        if (!blk[nb]) pb = p.total_blocks - 1;
        const int per_img = p.nby * p.nbx;
        const int n = pb / per_img;
        const int rem = pb - n * per_img;
        const int by = rem / p.nbx;
        const int bx = rem - by * p.nbx;
        img[nb] = n;
        by0[nb] = by * BH;
        bx0[nb] = bx * BW;
        const int yo = by0[nb] + ly, xo = bx0[nb] + lx;
        valid[nb] = blk[nb] && yo < p.h && xo < p.w;
        pixoff[nb] = valid[nb] ? (unsigned)(yo * p.w + xo) * 4u : MCQ_OOB;
        rsrc[nb] = mcq_make_rsrc(mcq_uniform_ptr(p.x + ((size_t)n * p.m + g) * (size_t)p.d * HW), group_bytes);
    }

    f32x4v A[PF];
    float B[PF][NP];
    const int tile_lo = blockIdx.z * q.tiles_per_z;
    const int tile_hi = tile_lo + q.tiles_per_z < p.ntile ? tile_lo + q.tiles_per_z : p.ntile;
    const float* wl = p.cbp + (((size_t)g * p.ntile + tile_lo) * p.Sp * 64 + lane) * 4;
    const f32x4v* c2l = reinterpret_cast<const f32x4v*>(p.c2p) + (size_t)g * (p.ntile + 1) * 64 + j;   // lane j of BOTH halves
    int ls = 0;
    unsigned soffL = 0;
    const unsigned step_bytes = 2u * (unsigned)HW * 4u;
    unsigned voffL[NP];
#pragma unroll
    for (int nb = 0; nb < NP; ++nb) voffL[nb] = valid[nb] ? pixoff[nb] + (unsigned)(hi * HW) * 4u : MCQ_OOB;

    auto issue = [&](int st) {
        A[st] = *reinterpret_cast<const f32x4v*>(wl);
        wl += 256;
#pragma unroll
        for (int nb = 0; nb < NP; ++nb) B[st][nb] = mcq_buffer_load(rsrc[nb], voffL[nb] + soffL);
        ++ls;
        soffL += step_bytes;
        if (ls == p.Sp) { ls = 0; soffL = 0; }
    };

    const float tmax = q.raw ? 1.0f : fmaxf(q.temperature[g], q.bound);
#pragma unroll
    for (int st = 0; st < PF; ++st) issue(st);

    // |x_v|^2 per vector (lane j), then broadcast into accumulator ROW layout: D[i][*] = x2[i].  Not needed for the raw inner products;
    // both vector blocks' loads of a batch are in flight together (round 6: the 2 x d / 16 dependent round trips of the first form were
    // 8-16 us in front of every launch at d = 64), behind the operand rings' first loads.
    f32x16 x2d[NP];
#pragma unroll
    for (int nb = 0; nb < NP; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) x2d[nb][r] = 0.0f;
    if (!q.raw) {
        const float bone = hi == 0 ? 1.0f : 0.0f;
        float s[NP];
#pragma unroll
        for (int nb = 0; nb < NP; ++nb) s[nb] = 0.0f;
        for (int c0 = 0; c0 < p.d; c0 += 16) {          // sixteen independent loads per block and batch, additions in channel order
            float v[NP][16];
#pragma unroll
            for (int nb = 0; nb < NP; ++nb)
#pragma unroll
                for (int i = 0; i < 16; ++i) v[nb][i] = mcq_buffer_load(rsrc[nb], pixoff[nb] + (unsigned)(c0 + i) * (unsigned)HW * 4u);
#pragma unroll
            for (int nb = 0; nb < NP; ++nb)
#pragma unroll
                for (int i = 0; i < 16; ++i) s[nb] = s[nb] + v[nb][i] * v[nb][i];
        }
#pragma unroll
        for (int nb = 0; nb < NP; ++nb) {
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.0f;
            x2d[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(hi == 0 ? s[nb] : 0.0f, bone, z, 0, 0, 0);
        }
    }


    for (int tile = tile_lo; tile < tile_hi; ++tile) {
        const f32x4v c2t = c2l[(size_t)tile * 64];
        f32x16 acc[NP][NW];
#pragma unroll
        for (int nb = 0; nb < NP; ++nb)
#pragma unroll
            for (int wb = 0; wb < NW; ++wb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nb][wb][r] = 0.0f;

        for (int t = 0; t < p.Sp; t += PF) {
#pragma unroll
            for (int st = 0; st < PF; ++st) {
#pragma unroll
                for (int nb = 0; nb < NP; ++nb) {        // vector-block major; the slot is refilled right after its last use
#pragma unroll
                    for (int wb = 0; wb < NW; ++wb)      // A operand = latent vectors (rows), B operand = codewords (cols)
                        acc[nb][wb] = __builtin_amdgcn_mfma_f32_32x32x2f32(B[st][nb], A[st][wb], acc[nb][wb], 0, 0, 0);
                    B[st][nb] = mcq_buffer_load(rsrc[nb], voffL[nb] + soffL);
                }
                A[st] = *reinterpret_cast<const f32x4v*>(wl);
                wl += 256;
                ++ls;
                soffL += step_bytes;
                if (ls == p.Sp) { ls = 0; soffL = 0; }
                // keep the software pipeline as written (without the fence hipcc regroups the loads of the body and
                // waits for nearly all of them at the loop head: the ring then hides one step instead of PF)
                __builtin_amdgcn_sched_barrier(0);
            }
        }

#pragma unroll
        for (int nb = 0; nb < NP; ++nb) {
            if (!blk[nb]) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = mcq_drow(r, hi);                       // vector index inside the block
                const int yo = by0[nb] + (i >> p.bw_log2), xo = bx0[nb] + (i & (BW - 1));
                if (yo < p.h && xo < p.w) {
                    float* row = q.logits + ((((size_t)img[nb] * p.m + g) * p.h + yo) * p.w + xo) * (size_t)p.k;
#pragma unroll
                    for (int wb = 0; wb < NW; ++wb) {
                        const int word = tile * 128 + wb * 32 + j;
                        if (word < p.k) {
                            if (q.raw) {
                                row[word] = acc[nb][wb][r];
                            } else {
                                const float dist = __builtin_fmaf(-2.0f, acc[nb][wb][r], x2d[nb][r] + c2t[wb]);
                                row[word] = div_by_constant(-1.0f * dist, q.scale, q.rscale) * tmax;
                            }
                        }
                    }
                }
            }
        }
    }
}

// One wave per latent vector (n, g, pixel): logits row (in/out), uniform draws, frequency EMA row.
//   logit[c] += -1e9 where u_drop[c] ** p < freq[g, c]
//   code   = argmax_c logit[c]
//   index  = argmax_c (logit[c] + gumbel(u_gumbel[c]));  s = softmax(.)[index];  hot = (1 - s) + s
// u^e < f, decided like `powf(u, e) < f`.  The hardware pair v_log_f32 / v_exp_f32 gives u^e to ~1e-4 relative at the
// exponents in use (e <= 13, |log2 u| <= 24); only when that estimate is within 1e-3 of f does the libm powf decide (about
// one lane in 10^4).  The decision is therefore powf's own, bit for bit, at a fifth of its instruction count.
// ---- uniform draws made inside the kernels (round 4) ------------------------------------------------------------------------
// The reference draws two `torch.rand_like(logit)` tensors per level (quantizer.py:194-230): at the first level of a training
// step that is 2 x 134 MB written by the generator and read back here (and the Gumbel draw once more in backward).  With
// `rng_state` = {seed, offset} (two uint64 in device memory, so that a captured hipGraph sees a fresh offset on every replay)
// the same numbers are made where they are used: u(stream, i) = a 24-bit uniform in [0, 1) from two rounds of a 32-bit
// avalanche mixer over (seed, offset, stream, element index i = row * k + c) -- a counter-based generator: any kernel, any
// thread layout and the backward pass reproduce element i's draw from its index alone.  Not torch's Philox stream (no RNG-stream
// parity is promised by either side: the draws are i.i.d. uniforms); mcq_hash_uniform_f32 materialises them for tests.
struct RngState { uint32_t s0, s1, o0, o1; };
__device__ __forceinline__ RngState rng_load(const unsigned long long* st) {
    RngState r = {0u, 0u, 0u, 0u};
    if (st) {
        const unsigned long long seed = st[0], off = st[1];
        r.s0 = (uint32_t)seed; r.s1 = (uint32_t)(seed >> 32); r.o0 = (uint32_t)off; r.o1 = (uint32_t)(off >> 32);
    }
    return r;
}
__device__ __forceinline__ uint32_t rng_mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float rng_uniform(const RngState& r, uint32_t stream, size_t idx) {
    uint32_t h = rng_mix((uint32_t)idx ^ r.s0);
    h = rng_mix(h + (uint32_t)((unsigned long long)idx >> 32) * 0x9E3779B1u + r.s1 + r.o0 * 0x85EBCA77u + r.o1 * 0x27D4EB2Fu + stream * 0xC2B2AE3Du);
    return (float)(h >> 8) * 5.9604644775390625e-08f;            // k / 2^24, k in [0, 2^24): float32's own grid on [0, 1), like torch.rand
}
// element c of a row's draw: from the tensor the caller gave, or made here
#define MCQ_U(ptr, stream, c) ((ptr) ? (ptr)[c] : rng_uniform(rng, (stream), rowbase + (size_t)(c)))

__global__ void hash_uniform_kernel(const unsigned long long* __restrict__ st, uint32_t stream, float* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = rng_uniform(rng_load(st), stream, i);
}

__device__ __forceinline__ bool drop_decision(float u, float e, float f) {
    const float est = __builtin_amdgcn_exp2f(e * __builtin_amdgcn_logf(u));      // (u = 0: log2 = -inf, est = 0 like powf)
    const float tol = 1e-3f * fmaxf(est, f);
    if (fabsf(est - f) > tol) return est < f;
    return powf(u, e) < f;
}

// Gumbel noise -log(-log u) for u in [eps, 1 - eps].  The inner logarithm keeps libm's logf: near u = 1 -- where the LARGEST
// noise values, the ones that decide the arg-max, come from -- it is a difference of nearly equal numbers and needs the full
// relative accuracy.  The outer one sees t = -log u in [1.2e-7, 16] and runs on v_log_f32 (1 ulp in log2 t, i.e. an absolute
// error below 1e-6 on a noise value of order 1..16: the size of one float32 rounding of the perturbed logit itself).
__device__ __forceinline__ float gumbel_noise(float u) {
    return -(__builtin_amdgcn_logf(-logf(u)) * 0.693147182464599609375f);
}

// exp(x) for x <= 0 (soft-max terms relative to the row maximum) on v_exp_f32: x log2(e) is split like the SiLU of
// mcq_common.h (product error recovered with an fma and folded back through 2^t (1 + t_lo ln 2)), so the result carries
// ~1 ulp wherever it matters (x near 0) and terms below 2^-126 flush to 0 like expf's do once they leave float range.
__device__ __forceinline__ float exp_nonpos(float x) {
    const float c_hi = 1.44269502162933349609375f, c_lo = 1.925963033500971e-8f;
    float t = x * c_hi;
    const float tl = __builtin_fmaf(x, c_lo, __builtin_fmaf(x, c_hi, -t));
    t = fmaxf(t, -150.0f);
    const float e = __builtin_amdgcn_exp2f(t);
    return __builtin_fmaf(e, tl * 0.693147182464599609375f, e);
}

__global__ __launch_bounds__(256) void vq_gumbel_sample_kernel(float* __restrict__ logits, const float* __restrict__ u_drop,
                                                               const float* __restrict__ u_gumbel, const unsigned long long* __restrict__ rng_state,
                                                               const float* __restrict__ freq, const float* __restrict__ drop_exponent_ptr,
                                                               int64_t* __restrict__ codes, int64_t* __restrict__ index,
                                                               float* __restrict__ hot, unsigned long long* __restrict__ counts,
                                                               int rows, int m, int hw, int k) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int g = (row / hw) % m;
    float* lr = logits + (size_t)row * k;
    const size_t rowbase = (size_t)row * k;
    const RngState rng = rng_load(rng_state);
    const float* ud = u_drop ? u_drop + rowbase : nullptr;
    const float* ug = u_gumbel ? u_gumbel + rowbase : nullptr;
    const float* fr = freq + (size_t)g * k;
    const float eps = 1.1920928955078125e-07f;       // torch.finfo(float32).eps
    const float drop_exponent = drop_exponent_ptr[0];

    float best_l = -INFINITY, best_y = -INFINITY;
    int code = 0, idx = 0;
    for (int c = lane; c < k; c += 64) {
        float l = lr[c];
        if (drop_decision(MCQ_U(ud, 0u, c), drop_exponent, fr[c])) l = l + -1e9f;
        lr[c] = l;
        const float u = fminf(fmaxf(MCQ_U(ug, 1u, c), eps), 1.0f - eps);
        const float y = l + gumbel_noise(u);
        if (l > best_l) { best_l = l; code = c; }
        if (y > best_y) { best_y = y; idx = c; }
    }
    // wave-wide (value, first index) argmax for both
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ol = __shfl_xor(best_l, off); const int oc = __shfl_xor(code, off);
        if (ol > best_l || (ol == best_l && oc < code)) { best_l = ol; code = oc; }
        const float oy = __shfl_xor(best_y, off); const int oi = __shfl_xor(idx, off);
        if (oy > best_y || (oy == best_y && oi < idx)) { best_y = oy; idx = oi; }
    }
    // softmax denominator with the max subtracted (torch's softmax): s[index] = exp(0) / sum = 1 / sum
    float sum = 0.0f;
    for (int c = lane; c < k; c += 64) {
        const float u = fminf(fmaxf(MCQ_U(ug, 1u, c), eps), 1.0f - eps);
        sum += exp_nonpos((lr[c] + gumbel_noise(u)) - best_y);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    if (lane == 0) {
        const float s = 1.0f / sum;
        codes[row] = code;
        index[row] = idx;
        hot[row] = (1.0f - s) + s;                  // y_hard - y_soft + y_soft at the hot position
        if (counts) atomicAdd(counts + (size_t)g * k + code, 1ull);      // integer counts: the order of arrival does not matter
    }
}

__global__ void vq_dequant_soft_kernel(const int64_t* __restrict__ index, const float* __restrict__ hot,
                                       const float* __restrict__ cb, float* __restrict__ out, float* __restrict__ out_silu, int N, int m,
                                       int d, int hw, int k) {
    // one thread per OUTPUT element (pixel fastest: coalesced stores; the (index, hot) pair of a latent vector is re-read by its d
    // channel threads from cache) -- a thread per vector walking its d channels left 4096 threads for 8 x 2 x 16 x 16 vectors
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)N * m * d * hw;
    if (i >= total) return;
    const int pix = (int)(i % hw);
    const size_t t = i / hw;
    const int c = (int)(t % d);
    const size_t ng = t / d;                                    // n * m + g
    const int g = (int)(ng % m);
    const size_t v = ng * hw + pix;
    int64_t code = index[v];
    code = code < 0 ? 0 : (code >= k ? k - 1 : code);
    const float val = hot[v] * cb[((size_t)g * k + (size_t)code) * d + c];
    out[i] = val;
    if (out_silu) out_silu[i] = mcq_silu(val);                  // the consumer's act1(.), like mcq_vq_gather_f32's twin
}

// ---- backward of the soft assignment ----------------------------------------------------------------------------
// One wave per latent vector.  y = softmax(logit + gumbel) is recomputed from the saved (post-drop) logits and the
// gumbel draw; dS[k] = <dDeq_v, c_k> comes in and is overwritten by d dist[k]:
//   dz = y (dS - <y, dS>)            (straight-through: the gradient reaches the sample through y_soft only)
//   d dist = dz * (-Tb / sqrt(k)),   d Tb += sum_k dz[k] * logit[k] / Tb,   rowsum = sum_k d dist[k]
__global__ __launch_bounds__(256) void vq_softmax_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ u_gumbel,
                                                             const unsigned long long* __restrict__ rng_state,
                                                             float* __restrict__ ds, const float* __restrict__ temperature,
                                                             float bound, float scale, float* __restrict__ rowsum,
                                                             float* __restrict__ dtrow, const float* __restrict__ dlogits,
                                                             const float* __restrict__ raw_logits, int rows, int m, int hw, int k) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int g = (row / hw) % m;
    const float tb = fmaxf(temperature[g], bound);
    const float* lr = logits + (size_t)row * k;
    const size_t rowbase = (size_t)row * k;
    const RngState rng = rng_load(rng_state);
    const float* ug = u_gumbel ? u_gumbel + rowbase : nullptr;
    float* dr = ds + (size_t)row * k;
    // a gradient on the returned logits themselves (quantizer.py:232-239 hands back a graph-carrying tensor): it joins the
    // soft-max's gradient in front of `_logit`; the random drop's `+= -1e9` passes gradients through, so dropped entries
    // take part too and the temperature term needs their un-dropped values (raw_logits)
    const float* dl = dlogits ? dlogits + (size_t)row * k : nullptr;
    const float* rw = raw_logits ? raw_logits + (size_t)row * k : nullptr;
    const float eps = 1.1920928955078125e-07f;
    float mx = -INFINITY;
    for (int c = lane; c < k; c += 64) {
        const float u = fminf(fmaxf(MCQ_U(ug, 1u, c), eps), 1.0f - eps);
        mx = fmaxf(mx, lr[c] + gumbel_noise(u));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float sum = 0.0f, dot = 0.0f;
    for (int c = lane; c < k; c += 64) {
        const float u = fminf(fmaxf(MCQ_U(ug, 1u, c), eps), 1.0f - eps);
        const float e = exp_nonpos((lr[c] + gumbel_noise(u)) - mx);
        sum += e;
        dot += e * dr[c];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { sum += __shfl_xor(sum, off); dot += __shfl_xor(dot, off); }
    const float inv = 1.0f / sum;
    dot *= inv;                                   // <y, dS>
    float rs = 0.0f, dt = 0.0f;
    const float dscale = -tb / scale;
    for (int c = lane; c < k; c += 64) {
        const float u = fminf(fmaxf(MCQ_U(ug, 1u, c), eps), 1.0f - eps);
        const float y = exp_nonpos((lr[c] + gumbel_noise(u)) - mx) * inv;
        float dz = y * (dr[c] - dot);
        if (dl) {
            dz += dl[c];
            dt += dz * (rw[c] / tb);
        } else if (dz != 0.0f) dt += dz * (lr[c] / tb);  // dropped entries (logit = -1e9) have y = 0 exactly
        const float dd = dz * dscale;
        dr[c] = dd;
        rs += dd;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { rs += __shfl_xor(rs, off); dt += __shfl_xor(dt, off); }
    if (lane == 0) { rowsum[row] = rs; dtrow[row] = dt; }
}

// dx_v[j] = 2 x_v[j] rowsum_v - 2 sum_k ddist[v][k] c_k[j]; one wave per latent vector, lanes over j
__global__ __launch_bounds__(256) void vq_dx_kernel(const float* __restrict__ ddist, const float* __restrict__ rowsum,
                                                    const float* __restrict__ x, const float* __restrict__ cb, float* __restrict__ dx,
                                                    int rows, int m, int d, int hw, int k) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int pix = row % hw;
    const int ng = row / hw;                      // n * m + g
    const int g = ng % m;
    const float* dr = ddist + (size_t)row * k;
    const float* cg = cb + (size_t)g * k * d;
    const float rs = rowsum[row];
    for (int j = lane; j < d; j += 64) {
        float acc = 0.0f;
        for (int c = 0; c < k; ++c) acc = __builtin_fmaf(dr[c], cg[(size_t)c * d + j], acc);
        const size_t xi = ((size_t)ng * d + j) * hw + pix;
        dx[xi] = 2.0f * x[xi] * rs - 2.0f * acc;
    }
}

// dC[g][c][j] = 2 C[g][c][j] colsum_c - 2 sum_v ddist[v][c] x_v[j] + sum_{v: index_v = c} hot_v dDeq_v[j];
// one wave per codeword, lanes over j, vectors visited in order (deterministic, no atomics).  xt / dqt are the
// channel-major (NHWC) copies of the latent and of the incoming gradient.
__global__ __launch_bounds__(256) void vq_dc_kernel(const float* __restrict__ ddist, const float* __restrict__ xt,
                                                    const float* __restrict__ dqt, const int64_t* __restrict__ index,
                                                    const float* __restrict__ hot, const float* __restrict__ cb,
                                                    float* __restrict__ dcb, int N, int m, int d, int hw, int k) {
    const int lane = threadIdx.x & 63;
    const int word = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int g = blockIdx.y;
    if (word >= k) return;
    const int C = m * d;
    for (int j = lane; j < d; j += 64) {
        float acc = 0.0f, cs = 0.0f, sc = 0.0f;
        for (int n = 0; n < N; ++n) {
            for (int p = 0; p < hw; ++p) {
                const size_t row = ((size_t)n * m + g) * hw + p;
                const float dd = ddist[row * k + word];
                const size_t t = ((size_t)n * hw + p) * C + (size_t)g * d + j;
                acc = __builtin_fmaf(dd, xt[t], acc);
                cs += dd;
                if (index[row] == word) sc = __builtin_fmaf(hot[row], dqt[t], sc);
            }
        }
        const size_t ci = ((size_t)g * k + word) * d + j;
        dcb[ci] = 2.0f * cb[ci] * cs - 2.0f * acc + sc;
    }
}


// ---- the two row kernels above with the row held in registers ---------------------------------------------------------------------
// T threads share one latent vector's row of k <= 8 T logits (T = 64: a wave, four rows per workgroup; 256; 1024: sixteen waves),
// eight elements per thread: every input is read ONCE, every logarithm / exponential is evaluated once -- the wave-per-row forms
// above walk the row two (forward) / three (backward) times and re-derive the Gumbel noise each time, which made them VALU-bound at
// k = 8192 (33.5 M elements x ~300 issue slots = the 241 / 290 us they took; the row's traffic alone is ~135 us).
constexpr int ROW_E = 8;

template <int T> struct RowShared {
    float v[4][T / 64];
    int i[2][T / 64];
};

// (value, first index) arg-max over the row's threads, for two quantities at once; every thread gets the result
template <int T>
__device__ __forceinline__ void row_argmax2(float& a, int& ai, float& b, int& bi, RowShared<T>& sm) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float oa = __shfl_xor(a, off); const int oi = __shfl_xor(ai, off);
        if (oa > a || (oa == a && oi < ai)) { a = oa; ai = oi; }
        const float ob = __shfl_xor(b, off); const int oj = __shfl_xor(bi, off);
        if (ob > b || (ob == b && oj < bi)) { b = ob; bi = oj; }
    }
    if constexpr (T > 64) {
        const int wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { sm.v[0][wave] = a; sm.i[0][wave] = ai; sm.v[1][wave] = b; sm.i[1][wave] = bi; }
        __syncthreads();
        a = sm.v[0][0]; ai = sm.i[0][0]; b = sm.v[1][0]; bi = sm.i[1][0];
#pragma unroll
        for (int w = 1; w < T / 64; ++w) {
            const float oa = sm.v[0][w]; const int oi = sm.i[0][w];
            if (oa > a || (oa == a && oi < ai)) { a = oa; ai = oi; }
            const float ob = sm.v[1][w]; const int oj = sm.i[1][w];
            if (ob > b || (ob == b && oj < bi)) { b = ob; bi = oj; }
        }
    }
}

// sums of two quantities over the row's threads (waves added in wave order); slots s0 / s0 + 1 of the shared scratch
template <int T>
__device__ __forceinline__ void row_sum2(float& a, float& b, RowShared<T>& sm, const int s0) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
    if constexpr (T > 64) {
        const int wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { sm.v[s0][wave] = a; sm.v[s0 + 1][wave] = b; }
        __syncthreads();
        a = sm.v[s0][0]; b = sm.v[s0 + 1][0];
#pragma unroll
        for (int w = 1; w < T / 64; ++w) { a += sm.v[s0][w]; b += sm.v[s0 + 1][w]; }
    }
}

template <int T>
__device__ __forceinline__ float row_max(float a, RowShared<T>& sm) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) a = fmaxf(a, __shfl_xor(a, off));
    if constexpr (T > 64) {
        const int wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) sm.v[0][wave] = a;
        __syncthreads();
        a = sm.v[0][0];
#pragma unroll
        for (int w = 1; w < T / 64; ++w) a = fmaxf(a, sm.v[0][w]);
    }
    return a;
}

// (two 1024-thread workgroups per CU = eight waves per SIMD = 64 registers: with one, the load / arithmetic / reduction phases of a row -- separated by
//  workgroup barriers -- have nothing to overlap with: 202 us = the row traffic plus the arithmetic, one after the other)
template <int T>
__global__ __launch_bounds__(T < 256 ? 256 : T, T == 1024 ? 8 : 1) void vq_gumbel_sample_row_kernel(float* __restrict__ logits, const float* __restrict__ u_drop,
                                                                                  const float* __restrict__ u_gumbel,
                                                                                  const unsigned long long* __restrict__ rng_state,
                                                                                  const float* __restrict__ freq,
                                                                                  const float* __restrict__ drop_exponent_ptr,
                                                                                  int64_t* __restrict__ codes, int64_t* __restrict__ index,
                                                                                  float* __restrict__ hot, unsigned long long* __restrict__ counts,
                                                                                  int rows, int m, int hw, int k) {
    __shared__ RowShared<T> sm;
    constexpr int RPW = T < 256 ? 256 / T : 1;
    const int tid = T < 256 ? (int)(threadIdx.x & (T - 1)) : (int)threadIdx.x;
    const int row = blockIdx.x * RPW + (T < 256 ? (int)(threadIdx.x / T) : 0);
    if (row >= rows) return;                                     // (T >= 256: the whole workgroup)
    const int g = (row / hw) % m;
    float* lr = logits + (size_t)row * k;
    const size_t rowbase = (size_t)row * k;
    const RngState rng = rng_load(rng_state);
    const float* ud = u_drop ? u_drop + rowbase : nullptr;
    const float* ug = u_gumbel ? u_gumbel + rowbase : nullptr;
    const float* fr = freq + (size_t)g * k;
    const float eps = 1.1920928955078125e-07f;
    const float drop_exponent = drop_exponent_ptr[0];
    float y[ROW_E];
    float best_l = -INFINITY, best_y = -INFINITY;
    int code = 0x7fffffff, idx = 0x7fffffff;
    // (two batches of four elements: sixteen loads in flight per thread, and the four inputs of an element die before the next
    //  batch arrives -- all eight at once need 40 live registers plus libm's, which does not fit the 64 of two workgroups per CU)
#pragma unroll
    for (int e0 = 0; e0 < ROW_E; e0 += 4) {
        float l[4], vd[4], vg[4], vf[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = tid + T * (e0 + e);
            const bool ok = c < k;
            l[e] = ok ? lr[c] : -INFINITY; vd[e] = ok ? MCQ_U(ud, 0u, c) : 1.0f; vg[e] = ok ? MCQ_U(ug, 1u, c) : 0.5f; vf[e] = ok ? fr[c] : 0.0f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = tid + T * (e0 + e);
            if (c < k) {
                if (drop_decision(vd[e], drop_exponent, vf[e])) l[e] = l[e] + -1e9f;
                lr[c] = l[e];
                const float u = fminf(fmaxf(vg[e], eps), 1.0f - eps);
                y[e0 + e] = l[e] + gumbel_noise(u);
                if (l[e] > best_l) { best_l = l[e]; code = c; }
                if (y[e0 + e] > best_y) { best_y = y[e0 + e]; idx = c; }
            } else
                y[e0 + e] = -INFINITY;
        }
    }
    // (a thread whose elements are all -inf / NaN keeps index INT_MAX: it loses every tie, like the serial scan that starts at 0)
    row_argmax2<T>(best_l, code, best_y, idx, sm);
    float sum = 0.0f, unused = 0.0f;
#pragma unroll
    for (int e = 0; e < ROW_E; ++e) sum += tid + T * e < k ? exp_nonpos(y[e] - best_y) : 0.0f;      // (exp_nonpos(-inf) is NaN)
    row_sum2<T>(sum, unused, sm, 2);
    if (tid == 0) {
        const float s = 1.0f / sum;
        const int cd = code == 0x7fffffff ? 0 : code;
        codes[row] = cd;
        index[row] = idx == 0x7fffffff ? 0 : idx;
        hot[row] = (1.0f - s) + s;
        // the level's code histogram (entropyCoder.py:33-35: one-hot codes summed over images and pixels), added where the code is made
        if (counts) atomicAdd(counts + (size_t)g * k + cd, 1ull);
    }
}

template <int T>
__global__ __launch_bounds__(T < 256 ? 256 : T, T == 1024 ? 8 : 1) void vq_softmax_bwd_row_kernel(const float* __restrict__ logits, const float* __restrict__ u_gumbel,
                                                                                const unsigned long long* __restrict__ rng_state,
                                                                                float* __restrict__ ds, const float* __restrict__ temperature,
                                                                                float bound, float scale, float* __restrict__ rowsum,
                                                                                float* __restrict__ dtrow, const float* __restrict__ dlogits,
                                                                                const float* __restrict__ raw_logits, int rows, int m, int hw, int k) {
    __shared__ RowShared<T> sm;
    constexpr int RPW = T < 256 ? 256 / T : 1;
    const int tid = T < 256 ? (int)(threadIdx.x & (T - 1)) : (int)threadIdx.x;
    const int row = blockIdx.x * RPW + (T < 256 ? (int)(threadIdx.x / T) : 0);
    if (row >= rows) return;
    const int g = (row / hw) % m;
    const float tb = fmaxf(temperature[g], bound);
    const float* lr = logits + (size_t)row * k;
    const size_t rowbase = (size_t)row * k;
    const RngState rng = rng_load(rng_state);
    const float* ug = u_gumbel ? u_gumbel + rowbase : nullptr;
    float* dr = ds + (size_t)row * k;
    const float* dl = dlogits ? dlogits + (size_t)row * k : nullptr;
    const float* rw = raw_logits ? raw_logits + (size_t)row * k : nullptr;
    const float eps = 1.1920928955078125e-07f;
    float l[ROW_E], y[ROW_E], dd[ROW_E];
#pragma unroll
    for (int e = 0; e < ROW_E; ++e) {
        const int c = tid + T * e;
        const bool ok = c < k;
        l[e] = ok ? lr[c] : -INFINITY; y[e] = ok ? MCQ_U(ug, 1u, c) : 0.5f; dd[e] = ok ? dr[c] : 0.0f;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < ROW_E; ++e) {
        const float u = fminf(fmaxf(y[e], eps), 1.0f - eps);
        y[e] = l[e] + gumbel_noise(u);                           // (-inf beyond the row)
        mx = fmaxf(mx, y[e]);
    }
    mx = row_max<T>(mx, sm);
    float sum = 0.0f, dot = 0.0f;
#pragma unroll
    for (int e = 0; e < ROW_E; ++e) {
        y[e] = tid + T * e < k ? exp_nonpos(y[e] - mx) : 0.0f;
        sum += y[e];
        dot += y[e] * dd[e];
    }
    row_sum2<T>(sum, dot, sm, 1);
    const float inv = 1.0f / sum;
    dot *= inv;
    float rs = 0.0f, dt = 0.0f;
    const float dscale = -tb / scale;
#pragma unroll
    for (int e = 0; e < ROW_E; ++e) {
        const int c = tid + T * e;
        if (c < k) {
            float dz = (y[e] * inv) * (dd[e] - dot);
            if (dl) {
                dz += dl[c];
                dt += dz * (rw[c] / tb);
            } else if (dz != 0.0f) dt += dz * (l[e] / tb);
            const float dv = dz * dscale;
            dr[c] = dv;
            rs += dv;
        }
    }
    if constexpr (T > 64) __syncthreads();                       // (slot 1 / 2 of the scratch are read by the sum above: reuse 2 / 3)
    row_sum2<T>(rs, dt, sm, 2);
    if (tid == 0) { rowsum[row] = rs; dtrow[row] = dt; }
}

}  // namespace

extern "C" int mcq_vq_logits_f32(const float* x, const float* cb_packed, const float* temperature, float bound, float* logits,
                                 int32_t N, int32_t m, int32_t d, int32_t h, int32_t w, int32_t k, void* stream) {
    if (!x || !cb_packed || !temperature || !logits || N <= 0 || m <= 0 || d <= 0 || h <= 0 || w <= 0 || k <= 0) return MCQ_EINVAL;
    if ((uint64_t)d * h * w * 4ull >= 0x80000000ull) return MCQ_ETOOLARGE;
    VqLogitK q;
    if (!vq_setup(q.v, x, cb_packed, N, m, d, h, w, k)) return MCQ_ETOOLARGE;
    q.v.codes = nullptr;
    q.temperature = temperature; q.bound = bound; q.logits = logits; q.raw = 0;
    // sqrt(k) as the reference computes it: math.sqrt (double) then used as a Python float in a float32 division
    q.scale = (float)sqrt((double)k);
    q.rscale = (float)(1.0 / (double)q.scale);
    const unsigned gx = (unsigned)(((q.v.total_blocks + VQ_NB - 1) / VQ_NB + 3) / 4);
    // (vector tiles x m) waves walk all codeword tiles otherwise: split the tiles over grid.z until ~2048 waves exist
    {
        const long long waves = (long long)gx * 4 * m;
        int zs = 1;
        while (zs < q.v.ntile && waves * zs < MCQ_LOGITS_WAVES) zs *= 2;
        q.tiles_per_z = (q.v.ntile + zs - 1) / zs;
        const unsigned gz = (unsigned)((q.v.ntile + q.tiles_per_z - 1) / q.tiles_per_z);
        hipLaunchKernelGGL(vq_logits_kernel, dim3(gx, (unsigned)m, gz), dim3(256), 0, (hipStream_t)stream, q);
    }
    return mcq_check_launch();
}

extern "C" int mcq_vq_inner_f32(const float* x, const float* cb_packed, float* out, int32_t N, int32_t m, int32_t d, int32_t h,
                                int32_t w, int32_t k, void* stream) {
    if (!x || !cb_packed || !out || N <= 0 || m <= 0 || d <= 0 || h <= 0 || w <= 0 || k <= 0) return MCQ_EINVAL;
    if ((uint64_t)d * h * w * 4ull >= 0x80000000ull) return MCQ_ETOOLARGE;
    VqLogitK q;
    if (!vq_setup(q.v, x, cb_packed, N, m, d, h, w, k)) return MCQ_ETOOLARGE;
    q.v.codes = nullptr;
    q.temperature = nullptr; q.bound = 0.0f; q.scale = 1.0f; q.rscale = 1.0f; q.logits = out; q.raw = 1;
    const unsigned gx = (unsigned)(((q.v.total_blocks + VQ_NB - 1) / VQ_NB + 3) / 4);
    // (vector tiles x m) waves walk all codeword tiles otherwise: split the tiles over grid.z until ~2048 waves exist
    {
        const long long waves = (long long)gx * 4 * m;
        int zs = 1;
        while (zs < q.v.ntile && waves * zs < MCQ_LOGITS_WAVES) zs *= 2;
        q.tiles_per_z = (q.v.ntile + zs - 1) / zs;
        const unsigned gz = (unsigned)((q.v.ntile + q.tiles_per_z - 1) / q.tiles_per_z);
        hipLaunchKernelGGL(vq_logits_kernel, dim3(gx, (unsigned)m, gz), dim3(256), 0, (hipStream_t)stream, q);
    }
    return mcq_check_launch();
}

extern "C" int mcq_hash_uniform_f32(const uint64_t* rng_state, uint32_t stream_id, float* out, int64_t n, void* stream) {
    if (!rng_state || !out || n <= 0) return MCQ_EINVAL;
    hipLaunchKernelGGL(hash_uniform_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const unsigned long long*>(rng_state), stream_id, out, (size_t)n);
    return mcq_check_launch();
}

extern "C" int mcq_vq_gumbel_sample_f32(float* logits, const float* u_drop, const float* u_gumbel, const uint64_t* rng_state_u64,
                                        const float* freq_ema,
                                        const float* drop_exponent, int64_t* codes, int64_t* sample_index, float* sample_hot,
                                        int64_t* code_counts, int32_t N, int32_t m, int32_t h, int32_t w, int32_t k, void* stream) {
    const unsigned long long* rng_state = reinterpret_cast<const unsigned long long*>(rng_state_u64);
    unsigned long long* counts = reinterpret_cast<unsigned long long*>(code_counts);
    if (!logits || !freq_ema || !drop_exponent || !codes || !sample_index || !sample_hot) return MCQ_EINVAL;
    if ((!u_drop || !u_gumbel) && !rng_state) return MCQ_EINVAL;        // every draw comes from a tensor or from the generator state
    if (N <= 0 || m <= 0 || h <= 0 || w <= 0 || k <= 0) return MCQ_EINVAL;
    const long long rows = (long long)N * m * h * w;
    if (rows > 0x7fffffffLL) return MCQ_ETOOLARGE;
    hipStream_t s = (hipStream_t)stream;
    if (k <= 64 * ROW_E)
        hipLaunchKernelGGL(vq_gumbel_sample_row_kernel<64>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, logits, u_drop, u_gumbel, rng_state,
                           freq_ema, drop_exponent, codes, sample_index, sample_hot, counts, (int)rows, m, h * w, k);
    else if (k <= 256 * ROW_E)
        hipLaunchKernelGGL(vq_gumbel_sample_row_kernel<256>, dim3((unsigned)rows), dim3(256), 0, s, logits, u_drop, u_gumbel, rng_state, freq_ema,
                           drop_exponent, codes, sample_index, sample_hot, counts, (int)rows, m, h * w, k);
    else if (k <= 1024 * ROW_E)
        hipLaunchKernelGGL(vq_gumbel_sample_row_kernel<1024>, dim3((unsigned)rows), dim3(1024), 0, s, logits, u_drop, u_gumbel, rng_state, freq_ema,
                           drop_exponent, codes, sample_index, sample_hot, counts, (int)rows, m, h * w, k);
    else
        hipLaunchKernelGGL(vq_gumbel_sample_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, logits,
                           u_drop, u_gumbel, rng_state, freq_ema, drop_exponent, codes, sample_index, sample_hot, counts, (int)rows, m, h * w, k);
    return mcq_check_launch();
}

extern "C" int mcq_vq_dequant_soft_f32(const int64_t* sample_index, const float* sample_hot, const float* codebook, float* out,
                                       float* out_silu, int32_t N, int32_t m, int32_t d, int32_t h, int32_t w, int32_t k, void* stream) {
    if (!sample_index || !sample_hot || !codebook || !out || N <= 0 || m <= 0 || d <= 0 || h <= 0 || w <= 0 || k <= 0) return MCQ_EINVAL;
    const size_t total = (size_t)N * m * d * h * w;
    hipLaunchKernelGGL(vq_dequant_soft_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       sample_index, sample_hot, codebook, out, out_silu, N, m, d, h * w, k);
    return mcq_check_launch();
}

extern "C" int mcq_vq_softmax_bwd_f32(const float* logits, const float* u_gumbel, const uint64_t* rng_state_u64, float* ds_inout,
                                      const float* temperature,
                                      float bound, float* rowsum, float* dtrow, const float* dlogits, const float* raw_logits,
                                      int32_t N, int32_t m, int32_t h, int32_t w, int32_t k, void* stream) {
    const unsigned long long* rng_state = reinterpret_cast<const unsigned long long*>(rng_state_u64);
    if (!logits || (!u_gumbel && !rng_state) || !ds_inout || !temperature || !rowsum || !dtrow || N <= 0 || m <= 0 || h <= 0 || w <= 0 || k <= 0)
        return MCQ_EINVAL;
    if ((dlogits != nullptr) != (raw_logits != nullptr)) return MCQ_EINVAL;
    const long long rows = (long long)N * m * h * w;
    if (rows > 0x7fffffffLL) return MCQ_ETOOLARGE;
    hipStream_t s = (hipStream_t)stream;
    const float scale = (float)sqrt((double)k);
    if (k <= 64 * ROW_E)
        hipLaunchKernelGGL(vq_softmax_bwd_row_kernel<64>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, logits, u_gumbel, rng_state, ds_inout,
                           temperature, bound, scale, rowsum, dtrow, dlogits, raw_logits, (int)rows, m, h * w, k);
    else if (k <= 256 * ROW_E)
        hipLaunchKernelGGL(vq_softmax_bwd_row_kernel<256>, dim3((unsigned)rows), dim3(256), 0, s, logits, u_gumbel, rng_state, ds_inout, temperature,
                           bound, scale, rowsum, dtrow, dlogits, raw_logits, (int)rows, m, h * w, k);
    else if (k <= 1024 * ROW_E)
        hipLaunchKernelGGL(vq_softmax_bwd_row_kernel<1024>, dim3((unsigned)rows), dim3(1024), 0, s, logits, u_gumbel, rng_state, ds_inout, temperature,
                           bound, scale, rowsum, dtrow, dlogits, raw_logits, (int)rows, m, h * w, k);
    else
        hipLaunchKernelGGL(vq_softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, logits, u_gumbel, rng_state,
                           ds_inout, temperature, bound, scale, rowsum, dtrow, dlogits, raw_logits, (int)rows, m, h * w, k);
    return mcq_check_launch();
}

namespace {

// ---- tiled forms of the two kernels above for d <= 64 (the training geometry: d = 64) ---------------------------------
// Both put the channel j on the lane axis, so codebook rows / latent vectors are coalesced 256-byte loads, and take the
// dDist factors -- which are the same for every channel -- from wave-uniform (scalar) loads.  Enough waves are created
// (4 per workgroup, each a slice of the contraction) that their latencies overlap; the slices meet in LDS and are added
// in slice order, so both results are deterministic.

// dx: a workgroup owns DX_R consecutive latent vectors of one (image, group); wave w contracts codewords
// [w k/4, (w+1) k/4): DX_R FMAs per codebook-row load.
constexpr int DX_R = 8;
__global__ __launch_bounds__(256) void vq_dx_tiled_kernel(const float* __restrict__ ddist, const float* __restrict__ rowsum,
                                                          const float* __restrict__ x, const float* __restrict__ cb,
                                                          float* __restrict__ dx, int rows, int m, int d, int hw, int k) {
    __shared__ float part[3][DX_R][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int row0 = blockIdx.x * DX_R;                           // hw % DX_R == 0: the DX_R rows share (n, g)
    const int pix0 = row0 % hw;
    const int ng = row0 / hw;
    const int g = ng % m;
    const int kq = k >> 2;                                        // k % 16 == 0
    const float* dr = ddist + (size_t)row0 * k + (size_t)wave * kq;
    const float* cg = cb + ((size_t)g * k + (size_t)wave * kq) * d;
    const bool jok = lane < d;
    float acc[DX_R];
#pragma unroll
    for (int r = 0; r < DX_R; ++r) acc[r] = 0.0f;
    for (int c = 0; c < kq; c += 4) {
        float cv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) cv[u] = jok ? cg[(size_t)(c + u) * d + lane] : 0.0f;
#pragma unroll
        for (int r = 0; r < DX_R; ++r) {
            const f32x4v gq = *reinterpret_cast<const f32x4v*>(dr + (size_t)r * k + c);   // wave-uniform address
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[r] = __builtin_fmaf(gq[u], cv[u], acc[r]);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < DX_R; ++r) part[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0 && jok) {
#pragma unroll
        for (int r = 0; r < DX_R; ++r) {
            const float t = ((acc[r] + part[0][r][lane]) + part[1][r][lane]) + part[2][r][lane];
            const size_t xi = ((size_t)ng * d + lane) * hw + pix0 + r;
            dx[xi] = 2.0f * x[xi] * rowsum[row0 + r] - 2.0f * t;
        }
    }
}

// dcodebook: a workgroup owns DC_W consecutive codewords of one group; wave w contracts the latent vectors
// [w V/4, (w+1) V/4): per vector one load of its channel row and DC_W scalar dDist values -> DC_W FMAs per lane.  The
// sampled codeword of a vector (index / hot, the straight-through path of the soft dequantisation) is wave-uniform:
// its term is added by a rare uniform branch.
constexpr int DC_W = 16;
__global__ __launch_bounds__(256) void vq_dc_tiled_kernel(const float* __restrict__ ddist, const float* __restrict__ xt,
                                                          const float* __restrict__ dqt, const int64_t* __restrict__ index,
                                                          const float* __restrict__ hot, const float* __restrict__ cb,
                                                          float* __restrict__ dcb, int N, int m, int d, int hw, int k) {
    __shared__ float part[3][2 * DC_W + 1][64];                   // waves 1..3: acc[c], sc[c] per lane; row 2 DC_W: column sums
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int g = blockIdx.y;
    const int c0 = blockIdx.x * DC_W;                             // k % DC_W == 0
    const int C = m * d;
    const int V = N * hw;
    const int per = (V + 3) / 4;
    const int v0 = wave * per, v1 = v0 + per < V ? v0 + per : V;
    const bool jok = lane < d;
    float acc[DC_W], sc[DC_W], cs[DC_W];
#pragma unroll
    for (int u = 0; u < DC_W; ++u) { acc[u] = 0.0f; sc[u] = 0.0f; cs[u] = 0.0f; }
    for (int v = v0; v < v1; ++v) {
        const int n = v / hw, pp = v - n * hw;
        const size_t row = ((size_t)n * m + g) * hw + pp;
        const float xv = jok ? xt[((size_t)n * hw + pp) * C + (size_t)g * d + lane] : 0.0f;
        const float* gr = ddist + row * k + c0;                   // wave-uniform: DC_W consecutive floats
#pragma unroll
        for (int q = 0; q < DC_W / 4; ++q) {
            const f32x4v gq = *reinterpret_cast<const f32x4v*>(gr + 4 * q);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[4 * q + u] = __builtin_fmaf(gq[u], xv, acc[4 * q + u]);
                cs[4 * q + u] += gq[u];
            }
        }
        const int idx = (int)index[row] - c0;                     // wave-uniform
        if (idx >= 0 && idx < DC_W) {
            const float t = jok ? hot[row] * dqt[((size_t)n * hw + pp) * C + (size_t)g * d + lane] : 0.0f;
#pragma unroll
            for (int u = 0; u < DC_W; ++u) sc[u] = u == idx ? sc[u] + t : sc[u];
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int u = 0; u < DC_W; ++u) { part[wave - 1][u][lane] = acc[u]; part[wave - 1][DC_W + u][lane] = sc[u]; }
        if (lane < DC_W) {
            float mine = 0.0f;
#pragma unroll
            for (int u = 0; u < DC_W; ++u) mine = lane == u ? cs[u] : mine;
            part[wave - 1][2 * DC_W][lane] = mine;
        }
    }
    __syncthreads();
    if (wave == 0 && jok) {
#pragma unroll
        for (int u = 0; u < DC_W; ++u) {
            const float a = ((acc[u] + part[0][u][lane]) + part[1][u][lane]) + part[2][u][lane];
            const float s2 = ((sc[u] + part[0][DC_W + u][lane]) + part[1][DC_W + u][lane]) + part[2][DC_W + u][lane];
            const float col = ((cs[u] + part[0][2 * DC_W][u]) + part[1][2 * DC_W][u]) + part[2][2 * DC_W][u];
            const size_t ci = ((size_t)g * k + c0 + u) * d + lane;
            dcb[ci] = 2.0f * cb[ci] * col - 2.0f * a + s2;
        }
    }
}

}  // namespace

extern "C" int mcq_vq_soft_bwd_f32(const float* ddist, const float* rowsum, const float* x, const float* x_nhwc,
                                   const float* ddeq_nhwc, const int64_t* sample_index, const float* sample_hot,
                                   const float* codebook, float* dx, float* dcodebook, int32_t N, int32_t m, int32_t d, int32_t h,
                                   int32_t w, int32_t k, void* stream) {
    if (!ddist || !rowsum || !x || !x_nhwc || !ddeq_nhwc || !sample_index || !sample_hot || !codebook || !dx || !dcodebook)
        return MCQ_EINVAL;
    if (N <= 0 || m <= 0 || d <= 0 || h <= 0 || w <= 0 || k <= 0) return MCQ_EINVAL;
    const long long rows = (long long)N * m * h * w;
    if (rows > 0x7fffffffLL) return MCQ_ETOOLARGE;
    hipStream_t s = (hipStream_t)stream;
    const int hw = h * w;
    VqBwdK q;
    q.ddist = ddist; q.rowsum = rowsum; q.x = x; q.xt = x_nhwc; q.dqt = ddeq_nhwc; q.index = sample_index; q.hot = sample_hot;
    q.cb = codebook; q.dx = dx; q.dcb = dcodebook; q.N = N; q.m = m; q.d = d; q.hw = hw; q.k = k; q.rows = (int)rows;
    // the MFMA forms (vq_bwd_mfma.hip) wherever their tiling fits; the lane-per-channel forms below for the rest
    if (mcq_vq_dx_mfma_ok(q))
        mcq_vq_dx_mfma_launch(q, stream);
    else if (d <= 64 && hw % DX_R == 0 && k % 16 == 0)
        hipLaunchKernelGGL(vq_dx_tiled_kernel, dim3((unsigned)(rows / DX_R)), dim3(256), 0, s, ddist, rowsum, x, codebook, dx,
                           (int)rows, m, d, hw, k);
    else
        hipLaunchKernelGGL(vq_dx_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, ddist, rowsum, x, codebook, dx, (int)rows, m, d,
                           hw, k);
    if (mcq_vq_dc_mfma_ok(q))
        mcq_vq_dc_mfma_launch(q, stream);
    else if (d <= 64 && k % DC_W == 0)
        hipLaunchKernelGGL(vq_dc_tiled_kernel, dim3((unsigned)(k / DC_W), (unsigned)m), dim3(256), 0, s, ddist, x_nhwc, ddeq_nhwc,
                           sample_index, sample_hot, codebook, dcodebook, N, m, d, hw, k);
    else
        hipLaunchKernelGGL(vq_dc_kernel, dim3((unsigned)((k + 3) / 4), (unsigned)m), dim3(256), 0, s, ddist, x_nhwc, ddeq_nhwc, sample_index,
                           sample_hot, codebook, dcodebook, N, m, d, hw, k);
    return mcq_check_launch();
}
