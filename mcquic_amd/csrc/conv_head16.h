// 3x3 stride-1 convolution with at most 16 output channels on v_mfma_f32_16x16x4_f32 (included by conv_mfma.hip).
//
// The 32-row tile of the general kernel wastes 20 of its 32 rows on the 12-channel image head (128 -> 12 channels
// + PixelShuffle(2), mcquic/nn/convs.py:221-255 as used by mcquic/modules/compressor.py:163), the only full-resolution
// layer of the decoder.  The 16x16x4 form has the same rate (2048 flops / 32 cycles) with 16 rows and FOUR input
// channels per instruction:
//     A[i = l & 15][k = l >> 4] = W[co = i][ci = 4 s + k][tap]          (one float per lane and k-step)
//     B[k = l >> 4][j = l & 15] = x[ci = 4 s + k][pixel j of a 16-pixel row segment]
//     D[i = 4 (l >> 4) + r][j = l & 15], r = 0..3                        (4 accumulator registers)
// so lane group q = l >> 4 ends up with the four sub-pixels r of PixelShuffle channel q for pixel j: its stores are two
// float2 rows of the [N, Cout/4, 2H, 2W] output.  Same streaming structure as the general kernel: operands straight
// from L2 into a prefetch ring (weights and activations 9 k-steps ahead), out-of-image taps as out-of-range offsets,
// k walked channel-major / tap-inner; a wave owns H16_NB segments (64 pixels).
#pragma once

namespace {

struct Head16K {
    const float* x; const float* wp16; const float* bias; float* y;
    int N, Cin, H, W, Cout;
    int S4;                 // 4-channel groups: ceil(Cin / 4)
    int gpr;                // 16-pixel segments per image row
    long long total_groups; // N * H * gpr
    unsigned flags;
};

typedef float f32x4acc __attribute__((ext_vector_type(4)));
#ifndef MCQ_H16_PFB
#define MCQ_H16_PFB 9      // activations one 4-channel group ahead (18: 1267 vs 1188 us on the head conv -- fewer registers, 3 waves / SIMD)
#endif
#ifndef MCQ_H16_OCC
#define MCQ_H16_OCC 3
#endif
constexpr int H16_NB = 4, H16_PFA = 9, H16_PFB = MCQ_H16_PFB;
constexpr int H16_TAIL_STEPS = 16;     // zero steps after the last k-step (the weight ring over-reads H16_PFA of them)

template <int PRO>
__global__ __launch_bounds__(256, MCQ_H16_OCC) void conv_head16_kernel(Head16K p) {
    constexpr int NB = H16_NB, PFA = H16_PFA, PFB = H16_PFB, U = PFB;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // XCD-aware order as in conv_mfma_kernel: XCD k (workgroup id % 8) walks the k-th eighth of the row segments
    unsigned wg = blockIdx.x;
    {
        const unsigned nwg = gridDim.x, xcd = wg & 7u, slot = wg >> 3;
        wg = xcd * (nwg >> 3) + (xcd < (nwg & 7u) ? xcd : (nwg & 7u)) + slot;
    }
    const long long g0 = ((long long)wg * 4 + wave) * NB;
    if (g0 >= p.total_groups) return;
    const int k4 = lane >> 4, j = lane & 15;
    const int HW = p.H * p.W;
    const unsigned plane_bytes = (unsigned)p.Cin * (unsigned)HW * 4u;

    int img[NB], yy[NB], xx[NB];
    bool valid[NB];
    __amdgpu_buffer_rsrc_t rsrc[NB];
    const char* xb[NB];
    unsigned voff[NB][9];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        long long g = g0 + nb;
        const bool gv = g < p.total_groups;
        if (!gv) g = p.total_groups - 1;
        const int per_img = p.H * p.gpr;
        const int n = (int)(g / per_img);
        const int rem = (int)(g - (long long)n * per_img);
        const int y = rem / p.gpr;
        const int xg = rem - y * p.gpr;
        img[nb] = n; yy[nb] = y; xx[nb] = xg * 16 + j;
        valid[nb] = gv && xx[nb] < p.W;
        xb[nb] = reinterpret_cast<const char*>(mcq_uniform_ptr(p.x + (size_t)n * p.Cin * HW));
        rsrc[nb] = mcq_make_rsrc(xb[nb], plane_bytes);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yi = y + tap / 3 - 1, xi = xx[nb] + tap % 3 - 1;
            const bool inb = valid[nb] && yi >= 0 && yi < p.H && xi >= 0 && xi < p.W;
            voff[nb][tap] = inb ? (unsigned)(yi * p.W + xi + k4 * HW) * 4u : MCQ_OOB;
        }
    }

    float A[PFA];
    float B[PFB][NB];
    // no VALU instructions in the k-loop (as in conv_mfma_kernel): weights through a buffer load with a scalar offset,
    // the 4-channel-group offset of the activations in per-iteration descriptors built by the scalar unit
    const __amdgpu_buffer_rsrc_t wr = mcq_make_rsrc(p.wp16, 0x7fffffffu);
    const unsigned wlane = (unsigned)lane * 4u;
    unsigned wso = 0;
    const unsigned step_bytes = 4u * (unsigned)HW * 4u;      // one 4-channel group further
    unsigned soff = 0;
    f32x4acc acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4acc{0.0f, 0.0f, 0.0f, 0.0f};

#pragma unroll
    for (int st = 0; st < PFA; ++st) { A[st] = mcq_buffer_load_s(wr, wlane, wso); wso += 256; }
#pragma unroll
    for (int st = 0; st < PFB; ++st) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) B[st][nb] = mcq_buffer_load(rsrc[nb], voff[nb][st % 9] + (unsigned)(st / 9) * step_bytes);
    }

    static_assert(PFB % 9 == 0, "whole channel groups per body");
    for (int s = 0; s < p.S4; s += U / 9) {
        __amdgpu_buffer_rsrc_t rB[NB][U / 9];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int jg = 0; jg < U / 9; ++jg) {
                const unsigned off = soff + (unsigned)(PFB / 9 + jg) * step_bytes;
                const int left = (int)plane_bytes - (int)off;
                rB[nb][jg] = mcq_make_rsrc(xb[nb] + off, (unsigned)(left > 0 ? left : 0));
            }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (u > 0 && u % 9 == 0 && s + u / 9 >= p.S4) break;     // odd group count: the second half of the body is past it
            const int sa = u % PFA, sb = u % PFB;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                float v = B[sb][nb];
                if (PRO == PRO_SILU) v = mcq_silu(v);
                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[sa], v, acc[nb], 0, 0, 0);
            }
            A[sa] = mcq_buffer_load_s(wr, wlane, wso);
            wso += 256;
            const int tl = (u + PFB) % 9, ds = (u + PFB) / 9;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) B[sb][nb] = mcq_buffer_load(rB[nb][ds - PFB / 9], voff[nb][tl]);
            __builtin_amdgcn_sched_barrier(0);                      // keep the software pipeline as written
        }
        soff += (unsigned)(U / 9) * step_bytes;
    }

    // lane group q = k4 holds output channels 4 q + r (r = 0..3) of pixel j
    const int q = k4;
    float bias4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bias4[r] = (p.bias && 4 * q + r < p.Cout) ? p.bias[4 * q + r] : 0.0f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        if (!valid[nb]) continue;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[nb][r] + bias4[r];
        if (p.flags & MCQ_CONV_SHUFFLE2) {
            const int Co4 = p.Cout >> 2;
            if (q < Co4) {
                const size_t W2 = 2 * (size_t)p.W;
                float* o = p.y + (((size_t)img[nb] * Co4 + q) * (2 * (size_t)p.H) + 2 * (size_t)yy[nb]) * W2 + 2 * (size_t)xx[nb];
                *reinterpret_cast<f32x2v*>(o) = f32x2v{v[0], v[1]};
                *reinterpret_cast<f32x2v*>(o + W2) = f32x2v{v[2], v[3]};
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * q + r < p.Cout)
                    p.y[(((size_t)img[nb] * p.Cout + 4 * q + r) * p.H + yy[nb]) * p.W + xx[nb]] = v[r];
        }
    }
}

// `mode` selects which convolution the operand stream is for (w is always the layer's own OIHW weight [Co, Ci, ks, ks]):
//   0  the layer itself:                    W'[co][ci][tap] = w[co][ci][tap]                                  (Cout = Co, Cin = Ci)
//   1  its input gradient, stride 1:        W'[co][ci][tap] = w[ci][co][taps - 1 - tap]  (flip + transpose)    (Cout = Ci, Cin = Co)
//   2  its input gradient, stride 2 (3x3):  dX = PixelShuffle2(conv3x3(dY, W')), W'[4 c + 2 i + j][o][(ty+1, tx+1)] =
//      w[o][c][ky(i, ty)][kx(j, tx)] with (phase, offset) -> kernel index {(0,0): 1, (1,0): 2, (1,1): 0}, else 0   (Cout = 4 Ci, Cin = Co)
// so the backward pass packs straight from the parameter in one launch (no flip / transpose / scatter on the way).
__device__ __forceinline__ float pack_source(const float* __restrict__ w, int mode, int Co, int Ci, int ks, int co, int ci, int tap) {
    const int taps = ks * ks;
    if (mode == 0) return w[((size_t)co * Ci + ci) * taps + tap];
    if (mode == 1) return w[((size_t)ci * Ci + co) * taps + (taps - 1 - tap)];
    const int c = co >> 2, i = (co >> 1) & 1, j = co & 1;
    const int ty = tap / 3 - 1, tx = tap % 3 - 1;
    const int ky = i == 0 ? (ty == 0 ? 1 : -1) : (ty == 0 ? 2 : ty == 1 ? 0 : -1);
    const int kx = j == 0 ? (tx == 0 ? 1 : -1) : (tx == 0 ? 2 : tx == 1 ? 0 : -1);
    if (ky < 0 || kx < 0) return 0.0f;
    return w[((size_t)ci * Ci + c) * 9 + ky * 3 + kx];
}

// OIHW -> [S4 * 9 + tail][64 lanes]: lane l of k-step s * 9 + tap holds W'[co = l & 15][ci = 4 s + (l >> 4)][tap]
__global__ void pack_head16_kernel(const float* __restrict__ w, int Cout, int Cin, int S4, float* __restrict__ out, size_t total,
                                   int mode, int Co, int Ci, float scale) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int lane = (int)(i & 63);
    const size_t step = i >> 6;
    float v = 0.0f;
    if (step < (size_t)S4 * 9) {
        const int s = (int)(step / 9), tap = (int)(step - (size_t)s * 9);
        const int co = lane & 15, ci = 4 * s + (lane >> 4);
        if (co < Cout && ci < Cin) v = pack_source(w, mode, Co, Ci, 3, co, ci, tap) * scale;
    }
    out[i] = v;
}

inline bool head16_shape(int Cout, int ksize) { return Cout <= 16 && ksize == 3; }
inline size_t head16_floats(int Cin) { return ((size_t)((Cin + 3) / 4) * 9 + H16_TAIL_STEPS) * 64; }

}  // namespace
