// 3x3 stride-1 convolutions of launches too small to fill the chip, on v_mfma_f32_16x16x4_f32 (included by conv_mfma.hip).
//
// The general kernel's smallest tile is 32 output channels x 32 pixels, and when a launch has fewer than ~2048 such tiles it
// splits their k-steps over the 8 waves of ONE workgroup.  For the smallest maps of the network -- the 4x4 / 8x8 / 16x16 levels of
// a training step (8 x 256x256 crops), the 12x8 / 24x16 levels of a single 768x512 image -- that still leaves most of the GPU idle:
// two 8 x 128 x 4 x 4 problems are 64 workgroups, and the 576 MFMAs of a tile are 3.8 us of one CU's matrix pipes whatever the
// split (stamp probe, round 3: DESIGN.md section 3.3 -- of the ~9 us of such a launch the k-loop is 3.2-3.8 us, geometry 1.0,
// ring fill 0.6-1.5, reduction + epilogue 1.8).  Here the tile is 16 channels x 16 pixels: four times the workgroups, a quarter of
// the matrix work per CU.
//
//     A[i = l & 15][k = l >> 4] = W[co = 16 T + i][ci = 4 q + k][tap]      D[4 (l >> 4) + r][l & 15], r = 0..3
//     B[k = l >> 4][j = l & 15] = x[ci = 4 q + k][pixel j under the tap]    (pixels = 16 consecutive ones of the flattened N x H x W)
//
// A workgroup = one tile, its four waves = four slices of the input channels (Cin / 16 channel quads each, all nine taps): at
// Cin = 128 a wave's share is 72 MFMAs and 72 + 72 operand registers, so there is NO ring -- a wave requests its whole slice
// (18 sixteen-byte weight loads, 72 activation loads) and multiplies as the data arrives; wave 0 requests bias / residual / the
// silu' operand before anything else, so the epilogue finds them there.  The partial tiles (4 floats per lane) meet in LDS, wave 0
// adds them in slice order.  Weights: a fourth section of the general operand stream (conv_mfma.hip: pack_conv_weight_body),
// [Cout / 16][(Cin / 4) x 9 steps / 4][64 lanes][4]: a lane's four consecutive k-steps are one 16-byte load.
// Arithmetic and epilogue order are the general kernel's (bias, * silu'(.), + scale * residual, SiLU, twin); the summation order
// over k differs (four channels per step, slices of channels), like between any two of its tiles.
#pragma once

namespace {

struct T16Ptrs { const float* x; const float* wp; const float* bias; float* y; float* y2; const float* res; const float* mul; };
struct T16K {
    T16Ptrs p[MCQ_CONV_MAX_MULTI];
    int N, Cin, H, W, Cout;
    unsigned flags;
    float res_scale;
};

constexpr unsigned T16_FLAGS = MCQ_CONV_SILU_OUT | MCQ_CONV_RESIDUAL | MCQ_CONV_DUAL_SILU | MCQ_CONV_DSILU_MUL;
inline bool t16_shape(int Cout, int Cin, int ksize) { return ksize == 3 && Cout >= 32 && Cout % 16 == 0 && (Cin == 64 || Cin == 128); }
inline size_t t16_floats(int Cout, int Cin, int ksize) { return t16_shape(Cout, Cin, ksize) ? (size_t)(Cout / 16) * (size_t)(Cin / 4) * 9 * 64 : 0; }

template <int QS>           // channel quads per wave = Cin / 16
__global__ __launch_bounds__(256) void conv_t16_kernel(T16K k) {
    __shared__ f32x4v part[3][64];
    constexpr int STEPS = QS * 9;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int kq = lane >> 4, j = lane & 15;
    T16Ptrs P = k.p[0];
#pragma unroll
    for (int c = 1; c < MCQ_CONV_MAX_MULTI; ++c)
        if ((int)blockIdx.z == c) P = k.p[c];
    const int co0 = (int)blockIdx.y * 16;
    const int HW = k.H * k.W;
    const long long npix = (long long)k.N * HW;
    const long long pix = (long long)blockIdx.x * 16 + j;
    const bool valid = pix < npix;
    const int n = valid ? (int)(pix / HW) : 0;
    const int rem = valid ? (int)(pix - (long long)n * HW) : 0;
    const int y = rem / k.W, x = rem - y * k.W;

    // ---- the epilogue's inputs first (wave 0): lane (kq, j) finishes channels co0 + 4 kq + r of pixel j ----------------------------
    const unsigned fl = k.flags;
    const size_t obase = ((size_t)n * k.Cout + co0 + 4 * kq) * HW + rem;
    f32x4v bias4 = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
    float side_m[4], side_r[4];
    if (wave == 0) {
        if (P.bias) bias4 = *reinterpret_cast<const f32x4v*>(P.bias + co0 + 4 * kq);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            side_m[r] = (valid && (fl & MCQ_CONV_DSILU_MUL)) ? P.mul[obase + (size_t)r * HW] : 0.0f;
            side_r[r] = (valid && (fl & MCQ_CONV_RESIDUAL)) ? P.res[obase + (size_t)r * HW] : 0.0f;
        }
    }

    // ---- operands of this wave's slice: channels [16 QS wave / 4 ...): quads wave QS .. wave QS + QS - 1 ------------------------------
    const __amdgpu_buffer_rsrc_t xr = mcq_make_rsrc(P.x, (unsigned)((size_t)k.N * k.Cin * HW * 4u));
    unsigned voff[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int yi = y + tap / 3 - 1, xi = x + tap % 3 - 1;
        const bool inb = valid && yi >= 0 && yi < k.H && xi >= 0 && xi < k.W;
        voff[tap] = inb ? (unsigned)(((n * k.Cin + kq) * HW) + yi * k.W + xi) * 4u : MCQ_OOB;
    }
    const float* wt = P.wp + ((size_t)blockIdx.y * (size_t)(k.Cin / 4) * 9 + (size_t)wave * STEPS) * 64;
    const __amdgpu_buffer_rsrc_t wr = mcq_make_rsrc(mcq_uniform_ptr(wt), (unsigned)(STEPS * 64 * 4));
    f32x4v A[STEPS / 4];
    float B[STEPS];
    const unsigned qbytes = 4u * (unsigned)HW * 4u;                      // one channel quad further
    const unsigned q0 = (unsigned)(wave * QS) * qbytes;
#pragma unroll
    for (int g = 0; g < STEPS / 4; ++g) {
        A[g] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(wr, (int)(lane * 16), g * 1024, 0));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int st = 4 * g + u;
            B[st] = mcq_buffer_load_s(xr, voff[st % 9], q0 + (unsigned)(st / 9) * qbytes);
        }
        __builtin_amdgcn_sched_barrier(0);        // (requests in the order the MFMAs consume them: loads retire in order)
    }
    __builtin_amdgcn_sched_barrier(0);            // (every request goes out before the first MFMA: left alone, hipcc sinks each load to its use)
    f32x4v acc = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int st = 0; st < STEPS; ++st) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[st >> 2][st & 3], B[st], acc, 0, 0, 0);

    // ---- the four slices meet in LDS; wave 0 adds them in slice order and finishes the tile -----------------------------------------
    if (wave > 0) part[wave - 1][lane] = acc;
    __syncthreads();
    if (wave != 0 || !valid) return;
    acc = ((acc + part[0][lane]) + part[1][lane]) + part[2][lane];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float v = acc[r] + bias4[r];
        if (fl & MCQ_CONV_DSILU_MUL) v = v * mcq_dsilu(side_m[r]);
        if (fl & MCQ_CONV_RESIDUAL) v = v + k.res_scale * side_r[r];
        if (fl & MCQ_CONV_SILU_OUT) v = mcq_silu(v);
        P.y[obase + (size_t)r * HW] = v;
        if (fl & MCQ_CONV_DUAL_SILU) P.y2[obase + (size_t)r * HW] = mcq_silu(v);
    }
}

// 16 x 16 tiles the launch would have; the launcher takes this kernel up to T16_MAX_TILES = three per CU.  (Swept 256 ... 6144 on
// the batch-1 encode+decode and the training step: 6.04-6.06 ms / 23.5-23.6 ms anywhere in 256 ... 1024, 6.19 / 23.9 at 1536,
// 6.40 / 24.4-25.2 beyond -- a launch that fills the chip with 32-row tiles is better off with their operand reuse.)
constexpr long long T16_MAX_TILES = 768;
inline long long t16_tiles(long long npix, int Cout, int nprob) { return ((npix + 15) / 16) * (Cout / 16) * nprob; }

}  // namespace
