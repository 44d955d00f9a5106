// RESEARCH PROBE (round 4), NOT part of the library: measured and dropped.  To try it again: copy next to conv_t16.h, include it from
// conv_mfma.hip after conv_t16.h and route launches to conv_r16_kernel (commit 29f26f1 has the launcher glue and the test).
// Result on MI355X: in isolation (tools/microbench_conv.py under rocprofv3, L2-warm weights) two 8 x 128 x 16x16 problems take 14.8 us
// against 20-21 us for the split 32 x 32 tiles (res + twin), four problems 26.5 against 34 -- but inside the captured training step
// (tools/bench_train.py --graph) the step got SLOWER, 22.53 -> 22.74 ms: every wave streams the full 147 KB of its channel pair's
// weights, 150 MB through L2 per launch where the split tiles move 38 MB, and in the step those weights are HBM-cold.
//
// 3x3 stride-1 convolutions of launches with ONE TO A FEW 32 x 16 tiles per SIMD, on v_mfma_f32_16x16x4_f32 (included by conv_mfma.hip).
//
// Between the launches conv_t16.h takes (at most three 16 x 16 tiles per CU: the 4x4 / 8x8 maps of a training step) and the ones
// that fill the chip with 32-row tiles lie the 16x16 maps of a training step (8 crops: 2048 pixels x 128 channels per problem, two or
// four problems per launch) and the 12x8 level of a 32-image batch.  The general kernel runs them as 32 x 32 tiles whose k-steps are
// split over 2-4 waves of a workgroup: round-4 trace of the training step, 116 such launches at 16.5 us each against 7.7-15.4 us of
// MFMA time -- all waves in one round and in phase, 2.4 us of prologue, an LDS reduction of 2 us behind the slowest slice, and the
// owner wave alone in the epilogue.  A launch of that size has 512 outputs per SIMD: HALF a 32 x 32 tile.  So here a wave owns
// 32 output channels x 16 pixels over the WHOLE contraction: no split, no LDS, no barrier, every wave finishes its own outputs, and
// with two or four problems in the launch each SIMD holds 2-4 such waves in different phases.
//
//     A[i = l & 15][k = l >> 4] = W[co = 32 T + 16 t + i][ci = 4 q + k][tap]      D_t[4 (l >> 4) + r][l & 15], t = 0, 1, r = 0..3
//     B[k = l >> 4][j = l & 15] = x[ci = 4 q + k][pixel j under the tap]          (pixels = 16 consecutive ones of the flattened N x H x W)
//
// Operands: the fourth section of the packed stream (conv_t16.h's order, [Cout / 16][(Cin / 4) x 9 / 4][64 lanes][4]: a lane's four
// consecutive k-steps of one 16-row tile are one 16-byte load), two tiles per wave; activations as in conv_t16.h.  A group = four
// k-steps = 2 + 4 loads and 8 MFMAs; the ring runs R16_DEPTH groups ahead, and the loop body is 18 groups (72 k-steps = 8 channel
// quads x 9 taps) so that every ring slot, tap and channel-quad distance is a compile-time constant (Cin % 32 == 0).
// Arithmetic and epilogue order are the general kernel's (bias, * silu'(.), + scale * residual, SiLU, twin).
#pragma once

namespace {

constexpr int R16_DEPTH = 6;                     // ring depth in groups of four k-steps (divides the 18 groups of a loop body)

inline bool r16_shape(int Cout, int Cin, int ksize) { return ksize == 3 && Cout % 32 == 0 && Cin % 32 == 0 && t16_shape(Cout, Cin, ksize); }

__global__ __launch_bounds__(256) void conv_r16_kernel(T16K k) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int kq = lane >> 4, j = lane & 15;
    T16Ptrs P = k.p[0];
#pragma unroll
    for (int c = 1; c < MCQ_CONV_MAX_MULTI; ++c)
        if ((int)blockIdx.z == c) P = k.p[c];
    const int co0 = (int)blockIdx.y * 32;
    const int HW = k.H * k.W;
    const long long npix = (long long)k.N * HW;
    const long long pix0 = ((long long)blockIdx.x * 4 + wave) * 16;       // the wave's 16 pixels
    if (pix0 >= npix) return;                                             // (wave-uniform; there is no barrier in this kernel)
    const long long pix = pix0 + j;
    const bool valid = pix < npix;
    const int n = valid ? (int)(pix / HW) : 0;
    const int rem = valid ? (int)(pix - (long long)n * HW) : 0;
    const int y = rem / k.W, x = rem - y * k.W;

    // ---- the epilogue's inputs first: lane (kq, j) finishes channels co0 + 16 t + 4 kq + r of pixel j ------------------------------
    const unsigned fl = k.flags;
    size_t obase[2];
    f32x4v bias4[2];
    float side_m[2][4], side_r[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        obase[t] = ((size_t)n * k.Cout + co0 + 16 * t + 4 * kq) * HW + rem;
        bias4[t] = P.bias ? *reinterpret_cast<const f32x4v*>(P.bias + co0 + 16 * t + 4 * kq) : f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            side_m[t][r] = (valid && (fl & MCQ_CONV_DSILU_MUL)) ? P.mul[obase[t] + (size_t)r * HW] : 0.0f;
            side_r[t][r] = (valid && (fl & MCQ_CONV_RESIDUAL)) ? P.res[obase[t] + (size_t)r * HW] : 0.0f;
        }
    }

    // ---- operand streams ---------------------------------------------------------------------------------------------------------
    const __amdgpu_buffer_rsrc_t xr = mcq_make_rsrc(P.x, (unsigned)((size_t)k.N * k.Cin * HW * 4u));
    unsigned voff[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int yi = y + tap / 3 - 1, xi = x + tap % 3 - 1;
        const bool inb = valid && yi >= 0 && yi < k.H && xi >= 0 && xi < k.W;
        voff[tap] = inb ? (unsigned)(((n * k.Cin + kq) * HW) + yi * k.W + xi) * 4u : MCQ_OOB;
    }
    const int G = (k.Cin / 4) * 9 / 4;                                    // groups of four k-steps per 16-row tile
    const unsigned tile_bytes = (unsigned)G * 1024u;
    const float* wt = P.wp + (size_t)(2 * blockIdx.y) * (size_t)G * 256;
    const __amdgpu_buffer_rsrc_t wr = mcq_make_rsrc(mcq_uniform_ptr(wt), 2u * tile_bytes);     // (reads past the two tiles return 0)
    const unsigned wlane = (unsigned)lane * 16u;
    const unsigned qbytes = 4u * (unsigned)HW * 4u;                      // one channel quad further

    f32x4v A[R16_DEPTH][2];
    float B[R16_DEPTH][4];
    unsigned wso = 0;                                                     // byte offset of the next group to request (tile 0)
    unsigned qso = 0;                                                     // byte offset of the channel quad the current body starts at
    // group gl of a body (gl may run past 17: the ring looks ahead into the next body): k-steps 4 gl + u, channel quad (4 gl + u) / 9
    // of the body, tap (4 gl + u) % 9 -- compile-time numbers once the body is unrolled
#define R16_ISSUE(slot, gl)                                                                                                         \
    do {                                                                                                                            \
        A[slot][0] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(wr, (int)wlane, (int)wso, 0));               \
        A[slot][1] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(wr, (int)wlane, (int)(wso + tile_bytes), 0)); \
        wso += 1024u;                                                                                                               \
        _Pragma("unroll") for (int u = 0; u < 4; ++u)                                                                               \
            B[slot][u] = mcq_buffer_load_s(xr, voff[(4 * (gl) + u) % 9], qso + (unsigned)((4 * (gl) + u) / 9) * qbytes);            \
    } while (0)

#pragma unroll
    for (int g = 0; g < R16_DEPTH; ++g) {
        R16_ISSUE(g, g);
        __builtin_amdgcn_sched_barrier(0);
    }
    f32x4v acc[2] = {f32x4v{0.0f, 0.0f, 0.0f, 0.0f}, f32x4v{0.0f, 0.0f, 0.0f, 0.0f}};
    const int bodies = k.Cin / 32;
    for (int it = 0; it < bodies; ++it) {
#pragma unroll
        for (int gl = 0; gl < 18; ++gl) {
            const int slot = gl % R16_DEPTH;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[slot][0][u], B[slot][u], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[slot][1][u], B[slot][u], acc[1], 0, 0, 0);
            }
            R16_ISSUE(slot, gl + R16_DEPTH);                              // (past the last body: out of range = 0, never used)
            __builtin_amdgcn_sched_barrier(0);
        }
        qso += 8u * qbytes;
    }
#undef R16_ISSUE

    if (!valid) return;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = acc[t][r] + bias4[t][r];
            if (fl & MCQ_CONV_DSILU_MUL) v = v * mcq_dsilu(side_m[t][r]);
            if (fl & MCQ_CONV_RESIDUAL) v = v + k.res_scale * side_r[t][r];
            if (fl & MCQ_CONV_SILU_OUT) v = mcq_silu(v);
            P.y[obase[t] + (size_t)r * HW] = v;
            if (fl & MCQ_CONV_DUAL_SILU) P.y2[obase[t] + (size_t)r * HW] = mcq_silu(v);
        }
}

// 32 x 16 tiles the launch would have; taken between R16_MIN_TILES (below: conv_t16.h or a split tile, which use more CUs) and
// R16_MAX_TILES (above: the 32-row tiles' operand reuse wins).  Both are run-time settable for sweeps (mcq_conv2d_r16_range).
long long g_r16_min_tiles = 1024, g_r16_max_tiles = 2048;
inline long long r16_tiles(long long npix, int Cout, int nprob) { return ((npix + 15) / 16) * (Cout / 32) * nprob; }

}  // namespace
