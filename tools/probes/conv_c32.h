// PROBE (round 6), not part of the library: it was wired into conv_launch (conv_mfma.hip) behind tile 0x320 / an automatic rule,
// measured, and taken out again -- see docs/experiments.md section 11.3 and profiles/r06_neon_c32.txt.  tools/probes/test_gpu_c32.py
// is the test it passed (49 cases bit-identical to the general kernel's unsplit 32 x 32 tile).
//
// 3x3 stride-1 convolutions between 32 channels (the layers of the reference's `Neon` model, channel = 32, at full resolution:
// mcquic/modules/compressor.py:181-241, mcquic/nn/blocks.py:162-200) -- a kernel designed for this width (included by conv_mfma.hip).
//
// The general kernel streams both MFMA operands from L2 through per-wave rings: right for 128 channels, where one 16-byte weight load
// feeds four MFMAs and a 576-step k-loop dwarfs a tile's prologue and epilogue.  At 32 -> 32 channels a tile is ONE 32-row band with a
// 144-step k-loop, and the layer is no longer compute-bound by a wide margin (72 FLOP per byte with a residual and a SiLU twin): its
// time on the general kernel is T(matrix) + T(HBM), the two do not overlap (4 x 512x512: 181 / 209 / 224 us with 2 / 3 / 4 tensor
// passes against 123 us of MFMAs -- every tile's first loads and last stores are exposed; profiles/r06_neon_c32.txt).  Here:
//   * the WHOLE filter bank lives in registers: 16 channel pairs x 9 taps = 144 k-steps = 144 VGPRs per lane (the 32-row section of
//     the packed operand stream, read once per workgroup), and the workgroups are PERSISTENT: they walk 8 x 32 pixel tiles (two rows per wave);
//   * activations reach the MFMAs through LDS, and reach LDS by DMA (`buffer_load ... lds`: no registers, zero padding = out-of-range
//     offsets): the 10 x 34 input patch of a tile in two halves of 16 channels, double-buffered -- while the waves multiply one half,
//     the other half (of this tile or the next) is in flight.  One workgroup barrier per half; the k-loop holds one LDS read and
//     1.5 MFMAs per step and no global load;
//   * two workgroups per CU (255 registers per lane; 2 x 45 KB of LDS), so one's epilogue runs under the other's MFMAs.
// Arithmetic: the same exact-fp32 MFMA and the general kernel's k-order (channel pair major, tap inner), bias / * silu'(.) /
// + scale * residual / SiLU / twin in its epilogue order -- bit-identical to its unsplit 32 x 32 tile.  No input prologue (the patch
// never passes through registers): SiLU-in launches bring the producer's SiLU twin, as everywhere on the inference path.
#pragma once

namespace {

struct C32K {
    const float* x; const float* wp32; const float* bias; float* y; float* y2; const float* res; const float* mul;
    int N, H, W;
    int tiles_x, tiles_y, ntiles;       // 8 x 32 pixel tiles per image row / column, in all
    unsigned flags; float res_scale;
};

constexpr unsigned C32_FLAGS = MCQ_CONV_SILU_OUT | MCQ_CONV_RESIDUAL | MCQ_CONV_DUAL_SILU | MCQ_CONV_DSILU_MUL;
#ifndef MCQ_C32_TH
#define MCQ_C32_TH 8
#endif
#ifndef MCQ_C32_OCC
#define MCQ_C32_OCC 2
#endif
constexpr int C32_TH = MCQ_C32_TH, C32_TW = 32;            // output tile: C32_TH / 4 pixel rows per wave
constexpr int C32_RP = C32_TH / 8;                          // row pairs per wave
constexpr int C32_PH = C32_TH + 2, C32_PW = C32_TW + 2;    // input patch
constexpr int C32_PLANE = C32_PH * C32_PW;                  // 340 floats per channel
constexpr int C32_LOADS = (C32_PLANE + 63) / 64;            // wave-wide DMA pieces per channel plane: 6, the last one 20 / 64 filled
constexpr int C32_WAVE = 4 * C32_PLANE + (C32_LOADS * 64 - C32_PLANE);     // a wave's four planes + what its last piece writes past them
constexpr int C32_HALF = 4 * C32_WAVE;                      // 16 channels: 5 616 floats (two halves: 44.9 KB per workgroup)

inline bool c32_shape(int Cout, int Cin, int ksize, int stride) { return Cout == 32 && Cin == 32 && ksize == 3 && stride == 1; }

// (two objects, not one array: hipcc then knows that a DMA into one half cannot alias the LDS reads of the other and does not wait for it)
__shared__ float c32_buf0[C32_HALF];
__shared__ float c32_buf1[C32_HALF];

__global__ __launch_bounds__(256, MCQ_C32_OCC) void conv_c32_kernel(C32K p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int hi = lane >> 5, j = lane & 31;
    const unsigned fl = p.flags;
    const int HW = p.H * p.W;
    const unsigned plane_bytes = 32u * (unsigned)HW * 4u;

    // ---- the filter bank: 144 k-steps, one float per lane each (lane (hi, i): W[co = i][ci = 2 s + hi][tap]) ----------------------
    float A[144];
    {
        const __amdgpu_buffer_rsrc_t wr = mcq_make_rsrc(p.wp32, 144u * 256u);
#pragma unroll
        for (int k = 0; k < 144; ++k) A[k] = mcq_buffer_load_s(wr, (unsigned)lane * 4u, (unsigned)k * 256u);
    }
    const __amdgpu_buffer_rsrc_t br = mcq_make_rsrc(p.bias ? p.bias : p.wp32, p.bias ? 32u * 4u : 0u);

    // XCD-aware tile order (see conv_mfma_kernel): XCD k walks the k-th eighth of the tiles, so neighbouring tiles share their halo in ITS L2
    const unsigned nwg = gridDim.x, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned per = ((unsigned)p.ntiles + 7u) >> 3, slots = (nwg + 7u - xcd) >> 3;        // tiles / workgroup slots of this XCD
    const unsigned t_end = min((xcd + 1u) * per, (unsigned)p.ntiles);
    const int per_img = p.tiles_x * p.tiles_y;

    // DMA of one half (16 channels) of tile `t`'s patch: wave w brings channels 16 half + 4 w .. + 3, ten pieces per plane; element e of a
    // plane = (row e / 34, col e % 34) of the patch whose corner is (y0 - 1, x0 - 1); e >= 612 and out-of-image taps read out of range = 0
    auto dma = [&](const unsigned t, const int half, float* buf) __attribute__((always_inline)) {
        const int n = (int)t / per_img, rem = (int)t - n * per_img;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const int y0 = ty * C32_TH, x0 = tx * C32_TW;
        const __amdgpu_buffer_rsrc_t xr = mcq_make_rsrc(reinterpret_cast<const char*>(p.x + (size_t)n * 32 * HW), plane_bytes);
        unsigned voff[C32_LOADS];
#pragma unroll
        for (int i = 0; i < C32_LOADS; ++i) {
            const int e = i * 64 + lane;
            const int row = (e * 241) >> 13, col = e - row * C32_PW;          // e / 34 for e < 640
            const int yy = y0 - 1 + row, xx = x0 - 1 + col;
            voff[i] = (e < C32_PLANE && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) ? (unsigned)(yy * p.W + xx) * 4u : MCQ_OOB;
        }
        float* dst = buf + wave * C32_WAVE;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < C32_LOADS; ++i)          // (in order: a plane's last piece writes zeros into the next plane's first 44 floats, which that plane's first piece then overwrites)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, dst + c * C32_PLANE + i * 64, 4, (int)voff[i], (int)((unsigned)((half * 16 + wave * 4 + c) * HW) * 4u), 0, 0);
    };
    // everything this wave has in flight has landed and every wave of the workgroup is here (LDS reads of the last half included)
    auto landed_and_met = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    f32x16 acc[C32_RP][2];
    // one half of the contraction for this wave's two pixel rows.  Per channel pair their nine taps read 3 columns x 4 patch rows =
    // 12 values for 18 MFMAs, requested one channel pair AHEAD; tap order = the general kernel's (dy major)
    auto multiply = [&](const float* buf, const int half) __attribute__((always_inline)) {
#pragma unroll
        for (int rp = 0; rp < C32_RP; ++rp) {
        // lane (hi, j): channel 2 s + hi of the half, column j + dx, patch rows of this row pair -- as an LDS byte address
        const unsigned l0 = (unsigned)reinterpret_cast<size_t>(buf + hi * C32_PLANE + j + (wave * 2 * C32_RP + 2 * rp) * C32_PW);
        f32x2v V[2][3][2];
        auto request = [&](const int s) __attribute__((always_inline)) {
            const unsigned base = l0 + (unsigned)(((s >> 1) * C32_WAVE + 2 * (s & 1) * C32_PLANE) * 4);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(V[s & 1][dx][0]) : "v"(base), "n"(dx), "n"(C32_PW + dx));
                asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(V[s & 1][dx][1]) : "v"(base), "n"(2 * C32_PW + dx), "n"(3 * C32_PW + dx));
            }
        };
        request(0);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // channel pair s has landed (requested 18 MFMAs ago)
            __builtin_amdgcn_sched_barrier(0);                        // (no MFMA of this pair may be scheduled above the wait)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3, dx = tap % 3;
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[rp][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(half * 8 + s) * 9 + tap], V[s & 1][dx][(b + dy) >> 1][(b + dy) & 1], acc[rp][b], 0, 0, 0);
                if (tap == 0 && s + 1 < 8) request(s + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        }
    };

    unsigned tile = xcd * per + slot;
    if (tile < t_end) dma(tile, 0, c32_buf0);
    for (; tile < t_end; tile += slots) {
#pragma unroll
        for (int rp = 0; rp < C32_RP; ++rp)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rp][b][r] = 0.0f;
        landed_and_met();                                    // channels 0 .. 15 are in buffer 0; nobody reads buffer 1 any more
        dma(tile, 1, c32_buf1);
        multiply(c32_buf0, 0);
        landed_and_met();                                    // channels 16 .. 31 are in buffer 1; nobody reads buffer 0 any more
        if (tile + slots < t_end) dma(tile + slots, 0, c32_buf0);
        multiply(c32_buf1, 1);

        // ---- epilogue: lane (hi, j) owns channels drow(r) + 4 hi of pixel (y0 + R, x0 + j) of its two rows ---------------------------
        const int n = (int)tile / per_img, rem = (int)tile - n * per_img;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const int y0 = ty * C32_TH, x0 = tx * C32_TW;
        const size_t oslab = (size_t)n * 32 * HW;
        const __amdgpu_buffer_rsrc_t yr = mcq_make_rsrc(mcq_uniform_ptr(p.y + oslab), plane_bytes);
#pragma unroll
        for (int rp = 0; rp < C32_RP; ++rp) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int yy = y0 + wave * 2 * C32_RP + 2 * rp + b, xx = x0 + j;
                const unsigned pvo = (yy < p.H && xx < p.W) ? ((unsigned)(yy * p.W + xx) + 4u * (unsigned)hi * (unsigned)HW) * 4u : MCQ_OOB;
                // (eight channels at a time: the filter bank and the four accumulator tiles leave ~40 registers for the side values)
                const __amdgpu_buffer_rsrc_t mr = mcq_make_rsrc(mcq_uniform_ptr(((fl & MCQ_CONV_DSILU_MUL) ? p.mul : p.y) + oslab), plane_bytes);
                const __amdgpu_buffer_rsrc_t rr = mcq_make_rsrc(mcq_uniform_ptr(((fl & MCQ_CONV_RESIDUAL) ? p.res : p.y) + oslab), plane_bytes);
                const __amdgpu_buffer_rsrc_t y2r = mcq_make_rsrc(mcq_uniform_ptr(((fl & MCQ_CONV_DUAL_SILU) ? p.y2 : p.y) + oslab), plane_bytes);
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    float v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = acc[rp][b][g * 8 + q] + mcq_buffer_load_s(br, (unsigned)hi * 16u, (unsigned)mcq_drow(g * 8 + q, 0) * 4u);
                    if (fl & MCQ_CONV_DSILU_MUL) {
                        float m[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) m[q] = mcq_buffer_load_s(mr, pvo, (unsigned)mcq_drow(g * 8 + q, 0) * (unsigned)HW * 4u);
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = v[q] * mcq_dsilu(m[q]);
                    }
                    if (fl & MCQ_CONV_RESIDUAL) {
                        float rv[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) rv[q] = mcq_buffer_load_s(rr, pvo, (unsigned)mcq_drow(g * 8 + q, 0) * (unsigned)HW * 4u);
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = v[q] + p.res_scale * rv[q];
                    }
                    if (fl & MCQ_CONV_SILU_OUT) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = mcq_silu(v[q]);
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) mcq_buffer_store_s(v[q], yr, pvo, (unsigned)mcq_drow(g * 8 + q, 0) * (unsigned)HW * 4u);
                    if (fl & MCQ_CONV_DUAL_SILU) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) mcq_buffer_store_s(mcq_silu(v[q]), y2r, pvo, (unsigned)mcq_drow(g * 8 + q, 0) * (unsigned)HW * 4u);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
}

}  // namespace
