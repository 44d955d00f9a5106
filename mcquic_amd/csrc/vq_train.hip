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
#include "../../include/mcquic_hip.h"
#include <math.h>

namespace {

struct VqLogitK {
    VqK v;
    const float* temperature;   // [m]
    float bound;
    float scale;                // sqrt(k)
    float* logits;              // [N, m, h, w, k]
};

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

    // |x_v|^2 per vector (lane j), then broadcast into accumulator ROW layout: D[i][*] = x2[i]
    f32x16 x2d[NP];
    const float bone = hi == 0 ? 1.0f : 0.0f;
#pragma unroll
    for (int nb = 0; nb < NP; ++nb) {
        float s = 0.0f;
        unsigned off = pixoff[nb];
        for (int c = 0; c < p.d; ++c) {
            const float v = mcq_buffer_load(rsrc[nb], off);
            s = s + v * v;
            off += (unsigned)HW * 4u;
        }
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.0f;
        x2d[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(hi == 0 ? s : 0.0f, bone, z, 0, 0, 0);
    }

    f32x4v A[PF];
    float B[PF][NP];
    const float* wl = p.cbp + ((size_t)g * p.ntile * p.Sp * 64 + lane) * 4;
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

    const float tmax = fmaxf(q.temperature[g], q.bound);
#pragma unroll
    for (int st = 0; st < PF; ++st) issue(st);

    for (int tile = 0; tile < p.ntile; ++tile) {
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
                for (int nb = 0; nb < NP; ++nb)
#pragma unroll
                    for (int wb = 0; wb < NW; ++wb)      // A operand = latent vectors (rows), B operand = codewords (cols)
                        acc[nb][wb] = __builtin_amdgcn_mfma_f32_32x32x2f32(B[st][nb], A[st][wb], acc[nb][wb], 0, 0, 0);
                issue(st);
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
                            const float dist = __builtin_fmaf(-2.0f, acc[nb][wb][r], x2d[nb][r] + c2t[wb]);
                            row[word] = ((-1.0f * dist) / q.scale) * tmax;
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
__global__ __launch_bounds__(256) void vq_gumbel_sample_kernel(float* __restrict__ logits, const float* __restrict__ u_drop,
                                                               const float* __restrict__ u_gumbel,
                                                               const float* __restrict__ freq, const float* __restrict__ drop_exponent_ptr,
                                                               int64_t* __restrict__ codes, int64_t* __restrict__ index,
                                                               float* __restrict__ hot, int rows, int m, int hw, int k) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int g = (row / hw) % m;
    float* lr = logits + (size_t)row * k;
    const float* ud = u_drop + (size_t)row * k;
    const float* ug = u_gumbel + (size_t)row * k;
    const float* fr = freq + (size_t)g * k;
    const float eps = 1.1920928955078125e-07f;       // torch.finfo(float32).eps
    const float drop_exponent = drop_exponent_ptr[0];

    float best_l = -INFINITY, best_y = -INFINITY;
    int code = 0, idx = 0;
    for (int c = lane; c < k; c += 64) {
        float l = lr[c];
        if (powf(ud[c], drop_exponent) < fr[c]) l = l + -1e9f;
        lr[c] = l;
        const float u = fminf(fmaxf(ug[c], eps), 1.0f - eps);
        const float y = l + (-logf(-logf(u)));
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
        const float u = fminf(fmaxf(ug[c], eps), 1.0f - eps);
        sum += expf((lr[c] + (-logf(-logf(u)))) - best_y);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    if (lane == 0) {
        const float s = 1.0f / sum;
        codes[row] = code;
        index[row] = idx;
        hot[row] = (1.0f - s) + s;                  // y_hard - y_soft + y_soft at the hot position
    }
}

__global__ void vq_dequant_soft_kernel(const int64_t* __restrict__ index, const float* __restrict__ hot,
                                       const float* __restrict__ cb, float* __restrict__ out, int N, int m, int d, int hw, int k) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)N * m * hw;
    if (i >= total) return;
    const int pix = (int)(i % hw);
    const size_t ng = i / hw;
    const int g = (int)(ng % m);
    const size_t n = ng / m;
    int64_t code = index[i];
    code = code < 0 ? 0 : (code >= k ? k - 1 : code);
    const float v = hot[i];
    const float* row = cb + ((size_t)g * k + (size_t)code) * d;
    float* o = out + ((n * m + g) * (size_t)d) * hw + pix;
    for (int c = 0; c < d; ++c) o[(size_t)c * hw] = v * row[c];
}

}  // namespace

extern "C" int mcq_vq_logits_f32(const float* x, const float* cb_packed, const float* temperature, float bound, float* logits,
                                 int32_t N, int32_t m, int32_t d, int32_t h, int32_t w, int32_t k, void* stream) {
    if (!x || !cb_packed || !temperature || !logits || N <= 0 || m <= 0 || d <= 0 || h <= 0 || w <= 0 || k <= 0) return MCQ_EINVAL;
    if ((uint64_t)d * h * w * 4ull >= 0x80000000ull) return MCQ_ETOOLARGE;
    VqLogitK q;
    if (!vq_setup(q.v, x, cb_packed, N, m, d, h, w, k)) return MCQ_ETOOLARGE;
    q.v.codes = nullptr;
    q.temperature = temperature; q.bound = bound; q.logits = logits;
    // sqrt(k) as the reference computes it: math.sqrt (double) then used as a Python float in a float32 division
    q.scale = (float)sqrt((double)k);
    const unsigned gx = (unsigned)(((q.v.total_blocks + VQ_NB - 1) / VQ_NB + 3) / 4);
    hipLaunchKernelGGL(vq_logits_kernel, dim3(gx, (unsigned)m), dim3(256), 0, (hipStream_t)stream, q);
    return mcq_check_launch();
}

extern "C" int mcq_vq_gumbel_sample_f32(float* logits, const float* u_drop, const float* u_gumbel, const float* freq_ema,
                                        const float* drop_exponent, int64_t* codes, int64_t* sample_index, float* sample_hot,
                                        int32_t N, int32_t m, int32_t h, int32_t w, int32_t k, void* stream) {
    if (!logits || !u_drop || !u_gumbel || !freq_ema || !drop_exponent || !codes || !sample_index || !sample_hot) return MCQ_EINVAL;
    if (N <= 0 || m <= 0 || h <= 0 || w <= 0 || k <= 0) return MCQ_EINVAL;
    const long long rows = (long long)N * m * h * w;
    if (rows > 0x7fffffffLL) return MCQ_ETOOLARGE;
    hipLaunchKernelGGL(vq_gumbel_sample_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, logits,
                       u_drop, u_gumbel, freq_ema, drop_exponent, codes, sample_index, sample_hot, (int)rows, m, h * w, k);
    return mcq_check_launch();
}

extern "C" int mcq_vq_dequant_soft_f32(const int64_t* sample_index, const float* sample_hot, const float* codebook, float* out,
                                       int32_t N, int32_t m, int32_t d, int32_t h, int32_t w, int32_t k, void* stream) {
    if (!sample_index || !sample_hot || !codebook || !out || N <= 0 || m <= 0 || d <= 0 || h <= 0 || w <= 0 || k <= 0) return MCQ_EINVAL;
    const size_t total = (size_t)N * m * h * w;
    hipLaunchKernelGGL(vq_dequant_soft_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       sample_index, sample_hot, codebook, out, N, m, d, h * w, k);
    return mcq_check_launch();
}
