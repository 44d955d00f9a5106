// Backward-pass kernels of the Compressor training step (BASELINE config #5) for gfx950.
//
// The reference gets these from torch.autograd over nn.Conv2d / F.conv2d / SiLU / GDN / sigmoid gates
// (mcquic/nn/convs.py, nn/gdn.py, nn/blocks.py); here:
//   * input gradients of every convolution reuse the forward MFMA kernel (conv_mfma.hip) with transformed weights
//     (flip + transpose; stride-2 and pixel-shuffle convs through the sub-pixel identity), so only the WEIGHT
//     gradient needs its own contraction;
//   * mcq_conv2d_wgrad_f32: dW[co][ci][tap] = sum over output pixels of dY[co][p] * X[ci][p + tap] -- a GEMM whose
//     reduction axis is the pixel axis.  Both operands are read channel-major (NHWC copies made by
//     mcq_nchw_to_nhwc_f32), which makes the MFMA operand loads runs of 32 consecutive floats exactly like the
//     forward kernel's; the pixel range is split over waves and the partial sums are reduced in a second,
//     deterministic pass (no atomics);
//   * small element-wise backward kernels (SiLU, gate, GDN) and reductions (bias / beta gradients).
#include <type_traits>

#include "mcq_common.h"
#include "../../include/mcquic_hip.h"

namespace {

// ---- NCHW -> NHWC (optionally squaring), 32 x 32 tiles through LDS --------------------------------------------
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ out, int C, int HW, int square) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 256 threads: 8 rows per pass
    const float* xi = x + (size_t)n * C * HW;
    float* oi = out + (size_t)n * C * HW;
#pragma unroll
    for (int r = 0; r < 32; r += 8) {
        const int c = c0 + ty + r, p = p0 + tx;
        float v = (c < C && p < HW) ? xi[(size_t)c * HW + p] : 0.0f;
        if (square) v = v * v;
        tile[ty + r][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 32; r += 8) {
        const int p = p0 + ty + r, c = c0 + tx;
        if (c < C && p < HW) oi[(size_t)p * C + c] = tile[tx][ty + r];
    }
}

// the two operands of one weight-gradient GEMM in a single launch: blockIdx.z < N transposes x, >= N transposes dy
__global__ void nchw_to_nhwc_pair_kernel(const float* __restrict__ x, float* __restrict__ xo, int Cx, int HWx, int square,
                                         const float* __restrict__ y, float* __restrict__ yo, int Cy, int HWy, int N) {
    __shared__ float tile[32][33];
    const bool second = (int)blockIdx.z >= N;
    const int n = second ? blockIdx.z - N : blockIdx.z;
    const int C = second ? Cy : Cx, HW = second ? HWy : HWx;
    const int c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    if (c0 >= C || p0 >= HW) return;                              // the grid covers the larger of the two shapes
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* xi = (second ? y : x) + (size_t)n * C * HW;
    float* oi = (second ? yo : xo) + (size_t)n * C * HW;
    const bool sq = !second && square;
#pragma unroll
    for (int r = 0; r < 32; r += 8) {
        const int c = c0 + ty + r, p = p0 + tx;
        float v = (c < C && p < HW) ? xi[(size_t)c * HW + p] : 0.0f;
        if (sq) v = v * v;
        tile[ty + r][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 32; r += 8) {
        const int p = p0 + ty + r, c = c0 + tx;
        if (c < C && p < HW) oi[(size_t)p * C + c] = tile[tx][ty + r];
    }
}

// ---- weight gradient -------------------------------------------------------------------------------------------
struct WgradK {
    const float* xt; const float* dyt; float* part; float* bias_part;
    int N, Cin, H, W, Cout, Ho, Wo, ks, stride;
    int splits;          // number of pixel ranges
    int chunk;           // output pixels per range (even)
    int ci_tiles;        // ceil(Cin / 64)
    long long P;         // N * Ho * Wo
};

#ifndef MCQ_WGRAD_WAVES
#define MCQ_WGRAD_WAVES 1024      // one wave per SIMD in all: fewer, longer pixel ranges beat 2048 / 4096 (57 -> 54 ms per
                                  // training step) and 512 (61 ms) -- less partial-sum traffic, fixed costs paid once
#endif
#ifndef MCQ_WGRAD_PF
#define MCQ_WGRAD_PF 8
#endif
constexpr int WG_MB = 4, WG_NB = 2, WG_PF = MCQ_WGRAD_PF;

// dW partials of one (pixel range, tap, 64-ci tile, 128-co tile) per wave: D[co][ci] += dy^T[p][co] * x^T[p + tap][ci],
// two output pixels per k-step (lane half hi takes pixel 2t + hi).  Both operands are NHWC copies, so a lane's 32
// channels are one 128-byte line.  Addressing is branch-free: the dy offset is linear in the pixel index and ends at the
// per-wave num_records (range end), channel / image-border predicates become the out-of-range marker, and the
// (n, y, x) walk of the input pixel uses selects.  The waves of tap 0 / ci-tile 0 also sum their dy operand over the
// pixels: that is the bias gradient of the same conv (bias_part[split][co]).
// EVEN (Wo even, the usual case): the two pixels of a k-step sit in one output row, so the (n, y, x) walk, the row part
// of the input offset and the image-border tests are wave-uniform and run on the scalar unit; a step then costs four
// vector instructions of addressing instead of ~55 (which, at one wave per SIMD, sat between the MFMAs: 38 % -> of peak).
template <bool EVEN>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(WgradK p) {
    constexpr int MB = WG_MB, NB = WG_NB, PF = WG_PF;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int split = blockIdx.x * 4 + wave;
    if (split >= p.splits) return;
    const int tap = blockIdx.y / p.ci_tiles;
    const int ci_base = (blockIdx.y - tap * p.ci_tiles) * (32 * NB);
    const int co_base = blockIdx.z * (32 * MB);
    const int hi = lane >> 5, j = lane & 31;
    const int pad = p.ks >> 1;
    const int dy = tap / p.ks, dx = tap - dy * p.ks;
    const long long p_begin = (long long)split * p.chunk;
    long long p_end = p_begin + p.chunk;
    if (p_end > p.P) p_end = p.P;
    const bool want_bias = p.bias_part != nullptr && blockIdx.y == 0;        // wave-uniform

    const __amdgpu_buffer_rsrc_t rx = mcq_make_rsrc(mcq_uniform_ptr(p.xt), (uint32_t)((size_t)p.N * p.H * p.W * p.Cin * 4));
    // EVEN walk: the wave-uniform part of an x address goes into the load's soffset (scalar unit); the descriptor starts
    // `pad` pixels before the tensor so that this part is never negative (the lanes that would read there are masked)
    const unsigned pad_bytes = (unsigned)((p.ks >> 1) * p.Cin * 4);
    const __amdgpu_buffer_rsrc_t rxs = mcq_make_rsrc(reinterpret_cast<const char*>(mcq_uniform_ptr(p.xt)) - pad_bytes,
                                                     (uint32_t)((size_t)p.N * p.H * p.W * p.Cin * 4) + pad_bytes);
    // dy^T rows [p_begin, p_end) of this wave: pixels past the range end are out of range = 0
    // (the last ranges can start at or past P when chunk was rounded up: empty descriptor)
    const long long d_begin = p_begin < p.P ? p_begin : p.P, d_len = p_end > p_begin ? p_end - p_begin : 0;
    const __amdgpu_buffer_rsrc_t rd = mcq_make_rsrc(mcq_uniform_ptr(p.dyt + (size_t)d_begin * p.Cout),
                                                    (uint32_t)((size_t)d_len * p.Cout * 4));

    // this lane's pixel: p_begin + hi, advancing by 2 per k-step
    const long long pp0 = p_begin + hi;
    const int HoWo = p.Ho * p.Wo;
    int n = (int)(pp0 / HoWo);
    const int rem = (int)(pp0 - (long long)n * HoWo);
    int yo = rem / p.Wo, xo = rem - yo * p.Wo;

    unsigned dbase[MB], xbase[NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
        dbase[mb] = co_base + 32 * mb + j < p.Cout ? (unsigned)(((size_t)hi * p.Cout + co_base + 32 * mb + j) * 4) : MCQ_OOB;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) xbase[nb] = ci_base + 32 * nb + j < p.Cin ? (unsigned)((ci_base + 32 * nb + j) * 4) : MCQ_OOB;
    unsigned dstep = 0;                                      // bytes from the range start to this k-step's pixel pair
    const unsigned dinc = 2u * (unsigned)p.Cout * 4u;

    // EVEN: wave-uniform walk of the pixel PAIR (p_begin is even, so is xo_u; lane half hi takes xo_u + hi, same row)
    int n_u = (int)(p_begin / HoWo);
    const int rem_u = (int)(p_begin - (long long)n_u * HoWo);
    int yo_u = rem_u / p.Wo, xo_u = rem_u - yo_u * p.Wo;
    unsigned xlane[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) xlane[nb] = xbase[nb] + (unsigned)(hi * p.stride * p.Cin) * 4u;   // (marker + small stays a marker)
    int xi0 = xo_u * p.stride + dx - pad;                         // input column of the pair's first pixel
    const int yi_first = yo_u * p.stride + dy - pad;
    bool row_ok = n_u < p.N && yi_first >= 0 && yi_first < p.H;
    unsigned row_base = row_ok ? (unsigned)(((n_u * p.H + yi_first) * p.W) * p.Cin * 4) : 0u;

    float A[PF][MB], B[PF][NB];
    auto issue = [&](int st) {
        if (EVEN) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) A[st][mb] = mcq_buffer_load_s(rd, dbase[mb], dstep);
            dstep += dinc;
            // row_ok / row_base only change when the walk wraps to a new row: kept in scalar registers, updated there
            const bool ok0 = row_ok && xi0 >= 0 && xi0 < p.W;
            const bool ok1 = row_ok && xi0 + p.stride >= 0 && xi0 + p.stride < p.W;
            const unsigned base_u = row_base + (unsigned)(xi0 * p.Cin * 4) + pad_bytes;   // >= 0 relative to rxs
            const bool ok = hi ? ok1 : ok0;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) B[st][nb] = mcq_buffer_load_s(rxs, ok ? xlane[nb] : MCQ_OOB, base_u);
            xo_u += 2;
            xi0 += 2 * p.stride;
            if (xo_u >= p.Wo) {
                xo_u = 0; ++yo_u;
                if (yo_u >= p.Ho) { yo_u = 0; ++n_u; }
                xi0 = dx - pad;
                const int yi = yo_u * p.stride + dy - pad;
                row_ok = n_u < p.N && yi >= 0 && yi < p.H;
                row_base = row_ok ? (unsigned)(((n_u * p.H + yi) * p.W) * p.Cin * 4) : 0u;
            }
            return;
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) A[st][mb] = mcq_buffer_load(rd, dbase[mb] + dstep);
        dstep += dinc;
        const int yi = yo * p.stride + dy - pad, xi = xo * p.stride + dx - pad;
        const bool inb = n < p.N && yi >= 0 && yi < p.H && xi >= 0 && xi < p.W;
        const unsigned xpix = inb ? (unsigned)(((n * p.H + yi) * p.W + xi) * p.Cin) * 4u : MCQ_OOB;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) B[st][nb] = mcq_buffer_load(rx, ((xpix | xbase[nb]) >> 31) ? MCQ_OOB : xpix + xbase[nb]);
        // advance two output pixels (select form: no divergent branches; Wo == 1 and Ho == 1 wrap twice)
        xo += 2;
        int w = xo >= p.Wo ? 1 : 0;
        xo -= w ? p.Wo : 0; yo += w;
        w = xo >= p.Wo ? 1 : 0;
        xo -= w ? p.Wo : 0; yo += w;
        w = yo >= p.Ho ? 1 : 0;
        yo -= w ? p.Ho : 0; n += w;
        w = yo >= p.Ho ? 1 : 0;
        yo -= w ? p.Ho : 0; n += w;
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;
    float bsum[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) bsum[mb] = 0.0f;

#pragma unroll
    for (int st = 0; st < PF; ++st) issue(st);
    const int steps = (p.chunk / 2 + PF - 1) / PF * PF;        // whole prefetch rounds; the tail loads are out of range = 0
    auto k_loop = [&](auto bias_tag) {
    constexpr bool BIAS = decltype(bias_tag)::value;
    for (int t = 0; t < steps; t += PF) {
#pragma unroll
        for (int st = 0; st < PF; ++st) {
            if (EVEN) {
                // ci-block major, every operand slot refilled right after its last use (the loads spread between the
                // MFMAs instead of queueing behind all eight); same addressing as issue()
                const bool ok0 = row_ok && xi0 >= 0 && xi0 < p.W;
                const bool ok1 = row_ok && xi0 + p.stride >= 0 && xi0 + p.stride < p.W;
                const unsigned base_u = row_base + (unsigned)(xi0 * p.Cin * 4) + pad_bytes;
                const bool ok = hi ? ok1 : ok0;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[st][mb], B[st][nb], acc[mb][nb], 0, 0, 0);
                    B[st][nb] = mcq_buffer_load_s(rxs, ok ? xlane[nb] : MCQ_OOB, base_u);
                }
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    if (BIAS) bsum[mb] = bsum[mb] + A[st][mb];       // only the waves of tap 0 / ci tile 0 (VALU work costs MFMA issue slots)
                    A[st][mb] = mcq_buffer_load_s(rd, dbase[mb], dstep);
                }
                dstep += dinc;
                xo_u += 2;
                xi0 += 2 * p.stride;
                if (xo_u >= p.Wo) {
                    xo_u = 0; ++yo_u;
                    if (yo_u >= p.Ho) { yo_u = 0; ++n_u; }
                    xi0 = dx - pad;
                    const int yi = yo_u * p.stride + dy - pad;
                    row_ok = n_u < p.N && yi >= 0 && yi < p.H;
                    row_base = row_ok ? (unsigned)(((n_u * p.H + yi) * p.W) * p.Cin * 4) : 0u;
                }
            } else {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[st][mb], B[st][nb], acc[mb][nb], 0, 0, 0);
                if (BIAS) {
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) bsum[mb] = bsum[mb] + A[st][mb];
                }
                issue(st);
            }
            __builtin_amdgcn_sched_barrier(0);                  // keep the software pipeline as written
        }
    }
    };
    if (want_bias) k_loop(std::true_type{});
    else k_loop(std::false_type{});

    // partial sums: part[split][tap][co][ci] (ci contiguous: 32 lanes = 128 B)
    const int taps = p.ks * p.ks;
    float* out = p.part + ((size_t)split * taps + tap) * (size_t)p.Cout * p.Cin;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int ci = ci_base + 32 * nb + j;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co_base + 32 * mb + mcq_drow(r, hi);
                if (co < p.Cout && ci < p.Cin) out[(size_t)co * p.Cin + ci] = acc[mb][nb][r];
            }
        }
    if (want_bias) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const float s = bsum[mb] + __shfl_xor(bsum[mb], 32);       // even + odd pixels
            const int co = co_base + 32 * mb + j;
            if (hi == 0 && co < p.Cout) p.bias_part[(size_t)split * p.Cout + co] = s;
        }
    }
}

// dW[co][ci][tap] = sum_split part[split][tap][co][ci]; db[co] = sum_split bias_part[split][co]   (fixed order: deterministic)
// A workgroup = 64 outputs x 4 slices of the split range: slice s sums its quarter of the splits with eight independent
// loads in flight, the four slice sums meet in LDS and are added in slice order.  (One thread per output walking all
// splits one dependent load at a time took 411 us for the 512 partials of a 1x1 conv's gradient.)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int splits, int taps, int Cout, int Cin,
                                    const float* __restrict__ bias_part, float* __restrict__ dbias) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * 64 + lane;                    // index into [tap][co][ci], then [co] of the bias
    const size_t per = (size_t)taps * Cout * Cin;
    const int per_slice = (splits + 3) / 4;
    const int s0 = slice * per_slice, s1 = s0 + per_slice < splits ? s0 + per_slice : splits;
    const bool is_bias = i >= per;
    const size_t co_b = i - per;
    const bool live = is_bias ? (dbias != nullptr && co_b < (size_t)Cout) : true;
    const float* src = is_bias ? bias_part + co_b : part + i;
    const size_t stride = is_bias ? (size_t)Cout : per;
    float s = 0.0f;
    if (live) {
        int sp = s0;
        for (; sp + 8 <= s1; sp += 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = src[(size_t)(sp + k) * stride];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
        for (; sp < s1; ++sp) s += src[(size_t)sp * stride];
    }
    red[slice][lane] = s;
    __syncthreads();
    if (slice != 0 || !live) return;
    s = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
    if (is_bias) { dbias[co_b] = s; return; }
    const int ci = (int)(i % Cin);
    const size_t r = i / Cin;
    const int co = (int)(r % Cout);
    const int tap = (int)(r / Cout);
    dw[((size_t)co * Cin + ci) * taps + tap] = s;
}

// out[c] = sum over n, pixels of x[n][c][p].  Stage 1: one workgroup per (channel, image-chunk) -> part[chunk][c];
// stage 2 (chunks == 1 skips it): fixed-order sum over the chunks.  Deterministic.
__global__ void channel_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int C, int HW, int nchunk) {
    __shared__ float red[256];
    const int c = blockIdx.x, chunk = blockIdx.y;
    const int per = (N + nchunk - 1) / nchunk;
    const int n0 = chunk * per, n1 = n0 + per < N ? n0 + per : N;
    float s = 0.0f;
    for (int n = n0; n < n1; ++n) {
        const float* row = x + ((size_t)n * C + c) * HW;
        for (int p = threadIdx.x; p < HW; p += 256) s += row[p];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[(size_t)chunk * C + c] = red[0];
}

__global__ void channel_sum_finish_kernel(const float* __restrict__ part, float* __restrict__ out, int C, int nchunk) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.0f;
    for (int k = 0; k < nchunk; ++k) s += part[(size_t)k * C + c];
    out[c] = s;
}

// ---- mean squared error (the distortion term, mcquic/loss/__init__.py:62) -------------------------------------------------------
// Two launches, no atomics and no memset: one double partial per workgroup, summed in a fixed order by one workgroup.  ATen's own
// large reductions zero their semaphores with hipMemsetAsync -- a memset NODE once captured, which ROCm 7.2's packet-captured graph
// launches replay wrongly after eager blit work (tools/probes/memset_node_probe.py): the captured step uses these instead.
constexpr int MSE_BLOCKS = 1024;

__global__ __launch_bounds__(256) void mse_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, double* __restrict__ part,
                                                          int64_t n) {
    __shared__ double red[256];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float d = b ? a[i] - b[i] : a[i];
        s += (double)(d * d);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void mse_finish_kernel(const double* __restrict__ part, int nblocks, double inv_n, float* __restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) s += part[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(red[0] * inv_n);
}

// x *= min(1, max_norm / (sqrt(sumsq) + eps)): torch.nn.utils.clip_grad_norm_'s scaling with the norm left on the device
__global__ __launch_bounds__(256) void clip_by_norm_kernel(float* __restrict__ x, const float* __restrict__ sumsq, float max_norm, float eps,
                                                           float* __restrict__ norm_out, int64_t n) {
    const float norm = sqrtf(sumsq[0]);
    const float coef = max_norm / (norm + eps);
    if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) norm_out[0] = norm;
    if (!(coef < 1.0f)) return;                              // (a NaN norm compares false: the gradients are left as they are)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] *= coef;
}

__global__ __launch_bounds__(256) void mse_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ dloss,
                                                      float scale, float* __restrict__ da, float* __restrict__ db, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const float g = (a[i] - b[i]) * (scale * dloss[0]);
        da[i] = g;
        if (db) db[i] = -g;
    }
}

// ---- Adam / AdamW over MANY tensors in one launch (the update of mcquic/train/trainer.py:283; torch.optim.Adam's arithmetic) ----------
// torch's fused Adam passes its tensor lists through kernel arguments (4 KB): 19 launches of ~91 us for this model's 666 tensors =
// 1.7 ms per step at 0.8 TB/s.  Here the lists live in device memory -- pointer tables [4][ntensors] (param, grad, exp_avg, exp_avg_sq),
// a block table (tensor, first element) of ADAM_CHUNK-element chunks -- so the whole model is ONE launch of ~13 k workgroups that moves
// 28 bytes per element once.  A one-thread kernel in front advances the step counter and derives the bias corrections on the device
// (double precision), so learning rate and step may be device scalars a captured graph re-reads on every replay.
constexpr int ADAM_CHUNK = 4096;
struct AdamScalars { float step_size; float inv_sqrt_bc2; float lr; float pad; };

__global__ void adam_prepare_kernel(float* __restrict__ step, const float* __restrict__ lr_dev, double lr_host, double beta1, double beta2,
                                    AdamScalars* __restrict__ sc) {
    if (blockIdx.x || threadIdx.x) return;
    const float t = step[0] + 1.0f;
    step[0] = t;
    const double lr = lr_dev ? (double)lr_dev[0] : lr_host;
    const double bc1 = 1.0 - pow(beta1, (double)t), bc2 = 1.0 - pow(beta2, (double)t);
    sc->step_size = (float)(lr / bc1);
    sc->inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    sc->lr = (float)lr;
    sc->pad = 0.0f;
}

__global__ __launch_bounds__(256) void adam_update_kernel(const unsigned long long* __restrict__ ptrs, int ntensors, const long long* __restrict__ numel,
                                                          const int* __restrict__ blk_tensor, const long long* __restrict__ blk_first,
                                                          const AdamScalars* __restrict__ sc, float omb1, float beta2, float omb2, float eps,
                                                          float weight_decay, int decoupled, int maximize) {
    const int t = blk_tensor[blockIdx.x];
    const long long first = blk_first[blockIdx.x];
    float* __restrict__ p = (float*)ptrs[t];
    const float* __restrict__ g = (const float*)ptrs[(size_t)ntensors + t];
    float* __restrict__ m = (float*)ptrs[2 * (size_t)ntensors + t];
    float* __restrict__ v = (float*)ptrs[3 * (size_t)ntensors + t];
    const long long n = numel[t];
    const long long end = first + ADAM_CHUNK < n ? first + ADAM_CHUNK : n;
    const float step_size = sc->step_size, inv_sqrt_bc2 = sc->inv_sqrt_bc2, lr = sc->lr;
    for (long long i = first + threadIdx.x; i < end; i += 256) {
        float gi = g[i], pi = p[i];
        if (maximize) gi = -gi;
        if (weight_decay != 0.0f) {
            if (decoupled) pi = pi * (1.0f - lr * weight_decay);     // AdamW: param.mul_(1 - lr * wd)
            else gi = gi + weight_decay * pi;                        // Adam: grad.add(param, alpha = wd)
        }
        float mi = m[i], vi = v[i];
        mi = mi + (gi - mi) * omb1;                                  // exp_avg.lerp_(grad, 1 - beta1)
        vi = vi * beta2 + omb2 * gi * gi;                            // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - step_size * (mi / denom);
    }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ void silu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = mcq_silu(x[i]);
}

__global__ void gate_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ x,
                                float* __restrict__ out, float* __restrict__ out_silu, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float v = a[i] * mcq_sigmoid(b[i]) + x[i];
        out[i] = v;
        if (out_silu) out_silu[i] = mcq_silu(v);               // the next block's act1(x), like MCQ_CONV_DUAL_SILU
    }
}

__global__ void axpby_kernel(const float* __restrict__ a, const float* __restrict__ b, float alpha, float beta,
                             float* __restrict__ out, float* __restrict__ out_silu, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float v = alpha * a[i] + beta * b[i];
        out[i] = v;
        if (out_silu) out_silu[i] = mcq_silu(v);               // the consumer's act1(.), like MCQ_CONV_DUAL_SILU
    }
}

// dx = dy * silu'(x),  silu'(x) = s (1 + x (1 - s)),  s = sigmoid(x)
// (+ `other`: the gradient that reaches x along a second path -- a strided / shuffle block's skip convolution reads x itself --,
//  added here instead of by an engine-issued add)
__global__ void silu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ other,
                                float* __restrict__ dx, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float s = sigmoidf_(x[i]);
        const float v = dy[i] * (s * (1.0f + x[i] * (1.0f - s)));
        dx[i] = other ? v + other[i] : v;
    }
}

// out = a * sigmoid(b) + x:  da = dout * s,  db = dout * a * s * (1 - s)
__global__ void gate_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ dout,
                                float* __restrict__ da, float* __restrict__ db, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float s = sigmoidf_(b[i]);
        const float g = dout[i];
        da[i] = g * s;
        db[i] = g * a[i] * s * (1.0f - s);
    }
}

// y = x * f(s): GDN f = s^-1/2, IGDN f = s^1/2.  dxd = dy * f(s);  ds = dy * x * f'(s)
__global__ void gdn_bwd_prep_kernel(const float* __restrict__ x, const float* __restrict__ s, const float* __restrict__ dy,
                                    int inverse, float* __restrict__ dxd, float* __restrict__ ds, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float sv = s[i], g = dy[i], xv = x[i];
        const float rs = 1.0f / sqrtf(sv);
        if (inverse) {                       // f = sqrt(s), f' = 1 / (2 sqrt(s))
            dxd[i] = g * sqrtf(sv);
            ds[i] = g * xv * (0.5f * rs);
        } else {                             // f = s^-1/2, f' = -1/2 s^-3/2
            dxd[i] = g * rs;
            ds[i] = g * xv * (-0.5f * rs * rs * rs);
        }
    }
}

// Backward of NonNegativeParametrizer (mcquic/nn/base.py:81-84 under LowerBound's gradient rule, :17-29):
//   folded = max(p, bound)^2 - eps;  g = 2 max(p, bound) dfolded;  dp = g where p >= bound or g < 0, else 0
__global__ void nonneg_reparam_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dfolded, float bound,
                                          float* __restrict__ dp, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float v = p[i];
        const float g = (2.0f * fmaxf(v, bound)) * dfolded[i];
        dp[i] = (v >= bound || g < 0.0f) ? g : 0.0f;
    }
}

// out[n][c*4 + i*2 + j][y][x] = in[n][c][2y + i][2x + j]   (inverse of nn.PixelShuffle(2))
__global__ void pixel_unshuffle2_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int C, int H, int W) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;     // over the INPUT [N, C, 2H, 2W]
    const size_t total = (size_t)N * C * 4 * H * W;
    if (i >= total) return;
    const int W2 = 2 * W, H2 = 2 * H;
    const int xx = (int)(i % W2);
    const size_t r1 = i / W2;
    const int yy = (int)(r1 % H2);
    const size_t nc = r1 / H2;
    const int c = (int)(nc % C);
    const size_t n = nc / C;
    const int sub = (yy & 1) * 2 + (xx & 1);
    out[(((n * C + c) * 4 + sub) * H + (yy >> 1)) * W + (xx >> 1)] = in[i];
}

}  // namespace

extern "C" int mcq_nchw_to_nhwc_f32(const float* x, float* out, int32_t N, int32_t C, int32_t HW, int32_t square, void* stream) {
    if (!x || !out || N <= 0 || C <= 0 || HW <= 0) return MCQ_EINVAL;
    const dim3 grid((unsigned)((HW + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)N);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, out, C, HW, square);
    return mcq_check_launch();
}

extern "C" int mcq_nchw_to_nhwc_pair_f32(const float* x, float* x_out, int32_t Cx, int32_t HWx, int32_t square_x, const float* y,
                                        float* y_out, int32_t Cy, int32_t HWy, int32_t N, void* stream) {
    if (!x || !x_out || !y || !y_out || N <= 0 || Cx <= 0 || Cy <= 0 || HWx <= 0 || HWy <= 0 || N > 32767) return MCQ_EINVAL;
    const int hw = HWx > HWy ? HWx : HWy, c = Cx > Cy ? Cx : Cy;
    const dim3 grid((unsigned)((hw + 31) / 32), (unsigned)((c + 31) / 32), (unsigned)(2 * N));
    hipLaunchKernelGGL(nchw_to_nhwc_pair_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, x_out, Cx, HWx, square_x, y, y_out, Cy,
                       HWy, N);
    return mcq_check_launch();
}

extern "C" size_t mcq_conv2d_wgrad_workspace_floats(int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t Cout, int32_t ksize,
                                                    int32_t stride) {
    if (N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || (ksize != 1 && ksize != 3) || (stride != 1 && stride != 2)) return 0;
    const int pad = ksize / 2;
    const long long Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
    const long long P = (long long)N * Ho * Wo;
    const int taps = ksize * ksize, ci_tiles = (Cin + 63) / 64, co_tiles = (Cout + 127) / 128;
    long long splits = MCQ_WGRAD_WAVES / ((long long)taps * ci_tiles * co_tiles);      // waves in all
    if (splits < 1) splits = 1;
    const long long max_splits = (P + 63) / 64;                 // at least 64 pixels per range
    if (splits > max_splits) splits = max_splits;
    return (size_t)splits * taps * Cout * Cin + (size_t)splits * Cout;      // dW partials + bias partials
}

extern "C" int mcq_conv2d_wgrad_f32(const float* x_nhwc, const float* dy_nhwc, float* dw, float* dbias, float* workspace, int32_t N,
                                    int32_t Cin, int32_t H, int32_t W, int32_t Cout, int32_t ksize, int32_t stride, void* stream) {
    if (!x_nhwc || !dy_nhwc || !dw || !workspace) return MCQ_EINVAL;
    const size_t wsf = mcq_conv2d_wgrad_workspace_floats(N, Cin, H, W, Cout, ksize, stride);
    if (wsf == 0) return MCQ_EINVAL;
    WgradK p;
    p.xt = x_nhwc; p.dyt = dy_nhwc; p.part = workspace;
    p.N = N; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout; p.ks = ksize; p.stride = stride;
    const int pad = ksize / 2;
    p.Ho = (H + 2 * pad - ksize) / stride + 1;
    p.Wo = (W + 2 * pad - ksize) / stride + 1;
    p.P = (long long)N * p.Ho * p.Wo;
    if ((uint64_t)N * H * W * Cin * 4ull >= 0x80000000ull || (uint64_t)p.P * Cout * 4ull >= 0x80000000ull) return MCQ_ETOOLARGE;
    const int taps = ksize * ksize;
    p.ci_tiles = (Cin + 63) / 64;
    const int co_tiles = (Cout + 127) / 128;
    p.splits = (int)(wsf / ((size_t)taps * Cout * Cin + (size_t)Cout));
    p.bias_part = dbias ? workspace + (size_t)p.splits * taps * Cout * Cin : nullptr;
    long long chunk = (p.P + p.splits - 1) / p.splits;
    chunk = (chunk + 1) & ~1LL;                                 // even: a k-step covers two pixels
    p.chunk = (int)chunk;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((p.splits + 3) / 4), (unsigned)(taps * p.ci_tiles), (unsigned)co_tiles);
    if (p.Wo % 2 == 0) hipLaunchKernelGGL(conv_wgrad_kernel<true>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(conv_wgrad_kernel<false>, grid, dim3(256), 0, s, p);
    const size_t per = (size_t)taps * Cout * Cin + (dbias ? (size_t)Cout : 0);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((per + 63) / 64)), dim3(256), 0, s, workspace, dw, p.splits, taps, Cout, Cin,
                       (const float*)p.bias_part, dbias);   // dW in OIHW order
    return mcq_check_launch();
}

extern "C" int mcq_channel_sum_f32(const float* x, float* out, float* workspace, int32_t N, int32_t C, int32_t HW, void* stream) {
    if (!x || !out || N <= 0 || C <= 0 || HW <= 0) return MCQ_EINVAL;
    // `workspace` (>= min(N, 16) * C floats) lets the images be summed by several workgroups per channel
    const int nchunk = workspace ? (N < 16 ? N : 16) : 1;
    hipStream_t s = (hipStream_t)stream;
    if (nchunk == 1) {
        hipLaunchKernelGGL(channel_sum_kernel, dim3((unsigned)C, 1), dim3(256), 0, s, x, out, N, C, HW, 1);
    } else {
        hipLaunchKernelGGL(channel_sum_kernel, dim3((unsigned)C, (unsigned)nchunk), dim3(256), 0, s, x, workspace, N, C, HW, nchunk);
        hipLaunchKernelGGL(channel_sum_finish_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, s, workspace, out, C, nchunk);
    }
    return mcq_check_launch();
}

static int mse_blocks(int64_t n) {
    const int64_t want = (n + 2047) / 2048;                  // >= 8 elements per lane before a workgroup is added
    return (int)(want < 1 ? 1 : want > MSE_BLOCKS ? MSE_BLOCKS : want);
}
extern "C" size_t mcq_mse_workspace_bytes(int64_t n) { return n > 0 ? (size_t)mse_blocks(n) * sizeof(double) : 0; }

extern "C" int mcq_mse_f32(const float* a, const float* b, float* out, void* workspace, int64_t n, void* stream) {
    if (!a || !b || !out || !workspace || n <= 0) return MCQ_EINVAL;
    const int nb = mse_blocks(n);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(mse_partial_kernel, dim3((unsigned)nb), dim3(256), 0, s, a, b, (double*)workspace, n);
    hipLaunchKernelGGL(mse_finish_kernel, dim3(1), dim3(256), 0, s, (const double*)workspace, nb, 1.0 / (double)n, out);
    return mcq_check_launch();
}

extern "C" int mcq_sumsq_f32(const float* x, float* out, void* workspace, int64_t n, void* stream) {
    if (!x || !out || !workspace || n <= 0) return MCQ_EINVAL;
    const int nb = mse_blocks(n);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(mse_partial_kernel, dim3((unsigned)nb), dim3(256), 0, s, x, (const float*)nullptr, (double*)workspace, n);
    hipLaunchKernelGGL(mse_finish_kernel, dim3(1), dim3(256), 0, s, (const double*)workspace, nb, 1.0, out);
    return mcq_check_launch();
}

extern "C" int mcq_clip_by_norm_f32(float* x, const float* sumsq, float max_norm, float eps, float* norm_out, int64_t n, void* stream) {
    if (!x || !sumsq || n <= 0 || !(max_norm > 0.0f) || !(eps >= 0.0f)) return MCQ_EINVAL;
    const int64_t want = (n + 1023) / 1024;
    hipLaunchKernelGGL(clip_by_norm_kernel, dim3((unsigned)(want < 1 ? 1 : want > 4096 ? 4096 : want)), dim3(256), 0, (hipStream_t)stream, x, sumsq,
                       max_norm, eps, norm_out, n);
    return mcq_check_launch();
}

extern "C" int mcq_mse_bwd_f32(const float* a, const float* b, const float* dloss, float* da, float* db, int64_t n, void* stream) {
    if (!a || !b || !dloss || !da || n <= 0) return MCQ_EINVAL;
    hipLaunchKernelGGL(mse_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, dloss, (float)(2.0 / (double)n),
                       da, db, n);
    return mcq_check_launch();
}

// many tensors -> their slices of one flat buffer in ONE launch: workgroup b copies chunk blk_first[b] of tensor blk_tensor[b]
// (device tables like the optimizer's: the tensor list does not travel as kernel arguments, 4 KB at a time)
__global__ __launch_bounds__(256) void gather_flat_kernel(const unsigned long long* __restrict__ src_ptrs, float* __restrict__ flat,
                                                          const long long* __restrict__ dst_off, const long long* __restrict__ numel,
                                                          const int* __restrict__ blk_tensor, const long long* __restrict__ blk_first) {
    const int t = blk_tensor[blockIdx.x];
    const long long first = blk_first[blockIdx.x];
    const float* __restrict__ src = (const float*)src_ptrs[t];
    float* __restrict__ dst = flat + dst_off[t];
    const long long n = numel[t];
    const long long end = first + ADAM_CHUNK < n ? first + ADAM_CHUNK : n;
    for (long long i = first + threadIdx.x; i < end; i += 256) dst[i] = src[i];
}

extern "C" int mcq_gather_flat_f32(const void* src_ptrs, float* flat, const int64_t* dst_offsets, const int64_t* numel, const int32_t* blk_tensor,
                                   const int64_t* blk_first, int32_t nblocks, void* stream) {
    if (!src_ptrs || !flat || !dst_offsets || !numel || !blk_tensor || !blk_first || nblocks <= 0) return MCQ_EINVAL;
    hipLaunchKernelGGL(gather_flat_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, (const unsigned long long*)src_ptrs, flat,
                       (const long long*)dst_offsets, (const long long*)numel, (const int*)blk_tensor, (const long long*)blk_first);
    return mcq_check_launch();
}

extern "C" int32_t mcq_adam_chunk(void) { return ADAM_CHUNK; }

extern "C" int mcq_adam_step_f32(const void* ptr_tables, int32_t ntensors, const int64_t* numel, const int32_t* blk_tensor, const int64_t* blk_first,
                                 int32_t nblocks, float* step, const float* lr_dev, double lr, double beta1, double beta2, double eps,
                                 double weight_decay, int32_t decoupled, int32_t maximize, void* scalars, void* stream) {
    if (!ptr_tables || !numel || !blk_tensor || !blk_first || !step || !scalars || ntensors <= 0 || nblocks <= 0) return MCQ_EINVAL;
    if (!(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0) || !(weight_decay >= 0.0)) return MCQ_EINVAL;
    if (!lr_dev && !(lr >= 0.0)) return MCQ_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(64), 0, s, step, lr_dev, lr, beta1, beta2, (AdamScalars*)scalars);
    hipLaunchKernelGGL(adam_update_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, (const unsigned long long*)ptr_tables, (int)ntensors,
                       (const long long*)numel, (const int*)blk_tensor, (const long long*)blk_first, (const AdamScalars*)scalars,
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)weight_decay, (int)decoupled, (int)maximize);
    // (1 - beta rounded from double, as torch passes `1 - beta2` to addcmul_: 1 - 0.999f would be 1.3e-5 off)
    return mcq_check_launch();
}

extern "C" int mcq_silu_f32(const float* x, float* y, int64_t n, void* stream) {
    if (!x || !y || n <= 0) return MCQ_EINVAL;
    hipLaunchKernelGGL(silu_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    return mcq_check_launch();
}

extern "C" int mcq_gate_f32(const float* a, const float* b, const float* x, float* out, float* out_silu, int64_t n, void* stream) {
    if (!a || !b || !x || !out || n <= 0) return MCQ_EINVAL;
    hipLaunchKernelGGL(gate_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, x, out, out_silu, n);
    return mcq_check_launch();
}

extern "C" int mcq_axpby_f32(const float* a, const float* b, float alpha, float beta, float* out, float* out_silu, int64_t n, void* stream) {
    if (!a || !b || !out || n <= 0) return MCQ_EINVAL;
    hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, alpha, beta, out, out_silu, n);
    return mcq_check_launch();
}

extern "C" int mcq_silu_bwd_f32(const float* x, const float* dy, const float* other, float* dx, int64_t n, void* stream) {
    if (!x || !dy || !dx || n <= 0) return MCQ_EINVAL;
    hipLaunchKernelGGL(silu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, dy, other, dx, n);
    return mcq_check_launch();
}

extern "C" int mcq_gate_bwd_f32(const float* a, const float* b, const float* dout, float* da, float* db, int64_t n, void* stream) {
    if (!a || !b || !dout || !da || !db || n <= 0) return MCQ_EINVAL;
    hipLaunchKernelGGL(gate_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, dout, da, db, n);
    return mcq_check_launch();
}

extern "C" int mcq_gdn_bwd_prep_f32(const float* x, const float* s, const float* dy, int32_t inverse, float* dx_direct, float* ds,
                                    int64_t n, void* stream) {
    if (!x || !s || !dy || !dx_direct || !ds || n <= 0) return MCQ_EINVAL;
    hipLaunchKernelGGL(gdn_bwd_prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, s, dy, inverse,
                       dx_direct, ds, n);
    return mcq_check_launch();
}

extern "C" int mcq_nonneg_reparam_bwd_f32(const float* p, const float* dfolded, float bound, float* dp, int64_t n, void* stream) {
    if (!p || !dfolded || !dp || n <= 0) return MCQ_EINVAL;
    hipLaunchKernelGGL(nonneg_reparam_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, dfolded, bound, dp, n);
    return mcq_check_launch();
}

extern "C" int mcq_pixel_unshuffle2_f32(const float* in, float* out, int32_t N, int32_t C, int32_t H, int32_t W, void* stream) {
    if (!in || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0) return MCQ_EINVAL;
    const size_t total = (size_t)N * C * 4 * H * W;
    hipLaunchKernelGGL(pixel_unshuffle2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out, N, C, H, W);
    return mcq_check_launch();
}
