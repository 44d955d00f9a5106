// RESEARCH PROBE, not part of libmcquic_hip: ONE layer shape -- 3x3 stride-1 convolution, Cin % 16 == 0,
// Cout % 128 == 0, W % 64 == 0 -- computed from float32 operands SPLIT into three bf16 planes each,
//     x = x0 + x1 + x2,  w = w0 + w1 + w2      (x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1): 24 mantissa bits in all)
// with the six product terms of order <= 2^-16 (x0w0, x0w1, x1w0, x0w2, x1w1, x2w0) on v_mfma_f32_32x32x16_bf16 and float32
// accumulation; the three dropped terms are <= 2^-24 |x||w| each, the size of one float32 product rounding.  bf16 x bf16 products
// are exact in float32; what differs from an fmaf chain is the order in which the hardware adds the 16 products of one MFMA.
// 6 MFMAs at 16x the fp32-MFMA rate = 2.67x its ceiling.  The question this probe answers on the GPU: how much of that survives the
// operand traffic (three planes in, channel-packed) and the split / re-pack pass in front of it.
//
// Layouts (channel-packed so that one 16-byte lane load is an MFMA operand):
//   activations  xs[plane][n][c / 8][h][w][8]  bf16        (mcq_probe_split_bf16x3: float32 NCHW -> three such planes)
//   weights      ws[plane][co / 128][c / 16][tap][band][lane][8] bf16, lane = 32 (k-half) + row: W[128 T + 32 band + row][16 s + 8 khalf + e][tap]
// MFMA operands: A (32 x 16): lane l holds row l % 32, k = 8 (l / 32) .. + 7;  B (16 x 32): lane l holds column l % 32, same k;
// D as the 32 x 32 float32 forms: register r of lane l = D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31].
// Wave tile 128 co x 64 px (two blocks of 32 pixels along x), one wave per SIMD (512 registers: operands double-buffered).
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ uint32_t bf16_rne(float x) {          // (finite inputs)
    uint32_t u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float bf16_to_f(uint32_t h) { return __uint_as_float(h << 16); }

// one thread per (n, channel group of 8, pixel): 8 strided float reads (coalesced across the pixels of a wave), 3 x 16-byte writes
__global__ void split_kernel(const float* __restrict__ x, u32x4* __restrict__ out, int N, int C, int HW, size_t plane_u4) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)N * (C / 8) * HW;
    if (idx >= total) return;
    const int pix = (int)(idx % HW);
    const size_t ncg = idx / HW;
    const int cg = (int)(ncg % (C / 8));
    const size_t n = ncg / (C / 8);
    const float* src = x + ((n * C + (size_t)cg * 8) * HW) + pix;
    uint32_t h[8], m[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = src[(size_t)e * HW];
        h[e] = bf16_rne(v);
        const float r1 = v - bf16_to_f(h[e]);
        m[e] = bf16_rne(r1);
        const float r2 = r1 - bf16_to_f(m[e]);
        l[e] = bf16_rne(r2);
    }
    out[idx] = u32x4{h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
    out[plane_u4 + idx] = u32x4{m[0] | (m[1] << 16), m[2] | (m[3] << 16), m[4] | (m[5] << 16), m[6] | (m[7] << 16)};
    out[2 * plane_u4 + idx] = u32x4{l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16)};
}

struct ConvP {
    const u32x4* xs; const u32x4* ws; const float* bias; float* y;
    int N, C, H, W, Cout, tiles;         // tiles = N * H * (W / 64)
    size_t x_plane_u4, w_plane_u4;       // 16-byte units per plane
    int terms;                           // 6 = the split product; 1 = x0 w0 only (plain bf16: the speed of light of this loop)
};

template <int TERMS>
__global__ __launch_bounds__(256, 1) void conv_bf16x3_kernel(ConvP p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= p.tiles) return;
    const int T = blockIdx.y;
    const int tiles_per_row = p.W / 64;
    const int n = tile / (p.H * tiles_per_row);
    const int rem = tile - n * (p.H * tiles_per_row);
    const int yrow = rem / tiles_per_row;
    const int x0 = (rem - yrow * tiles_per_row) * 64;
    const int j = lane & 31, kb = lane >> 5;
    const int S = p.C / 16;
    const int KT = S * 9;

    f32x16 acc[4][2];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;

    u32x4 A[2][4][3], B[2][2][3];
    const u32x4 zero = {0u, 0u, 0u, 0u};
    const size_t wbase = (size_t)T * S * 9 * 4 * 64 + lane;
    auto load = [&](const int st, const int it) __attribute__((always_inline)) {
        const int s = it / 9, tap = it - s * 9;
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const int yy = yrow + dy;
        const size_t wofs = wbase + (size_t)it * 4 * 64;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int pl = 0; pl < (TERMS == 1 ? 1 : 3); ++pl) A[st][mb][pl] = p.ws[pl * p.w_plane_u4 + wofs + (size_t)mb * 64];
        const size_t crow = ((size_t)n * (p.C / 8) + 2 * s + kb) * p.H;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int xx = x0 + 32 * nb + j + dx;
            const bool inb = yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
            const size_t o = (crow + (size_t)(inb ? yy : 0)) * p.W + (inb ? xx : 0);
#pragma unroll
            for (int pl = 0; pl < (TERMS == 1 ? 1 : 3); ++pl) {
                const u32x4 v = p.xs[pl * p.x_plane_u4 + o];
                B[st][nb][pl] = inb ? v : zero;
            }
        }
    };
    auto compute = [&](const int st) __attribute__((always_inline)) {
        // smallest terms first (x0 w2, x2 w0, x1 w1), then x0 w1, x1 w0, then x0 w0 -- (activation plane, weight plane)
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0};      // weight plane
        constexpr int PB[6] = {0, 2, 1, 0, 1, 0};      // activation plane
#pragma unroll
        for (int t = (TERMS == 1 ? 5 : 0); t < 6; ++t)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[st][mb][PA[t]]),
                                                                          __builtin_bit_cast(bf16x8, B[st][nb][PB[t]]), acc[mb][nb], 0, 0, 0);
    };
    load(0, 0);
    for (int it = 0; it < KT; it += 2) {               // (KT = 9 C / 16 is even whenever C % 32 == 0)
        if (it + 1 < KT) load(1, it + 1);
        compute(0);
        if (it + 2 < KT) load(0, it + 2);
        if (it + 1 < KT) compute(1);
    }

    // epilogue: + bias, float32 NCHW (rows of 32 consecutive pixels per register and lane half)
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = T * 128 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb;
            const float b = p.bias ? p.bias[co] : 0.0f;
            float* row = p.y + (((size_t)n * p.Cout + co) * p.H + yrow) * p.W + x0 + j;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) row[32 * nb] = acc[mb][nb][r] + b;
        }
}

// ---- v1: the weight operands of a k-step are the same for the four waves of a workgroup (one 128-row output-channel tile, four
// pixel tiles): each wave fetches ONE band's three planes from global memory (3 x 16 bytes per lane instead of 12) and parks them
// in LDS, double-buffered; after one workgroup barrier per k-step every wave reads all four bands back (12 ds_read_b128).  The
// global-memory path then carries 9 KB per wave and k-step instead of 18.  Everything is software-pipelined inside the single
// resident wave of a SIMD: global loads run two k-steps ahead, the LDS write one step ahead, and barrier + LDS reads sit in the
// middle of the 48 MFMAs of a step so that neither latency is exposed.
template <int TERMS>
__global__ __launch_bounds__(256, 1) void conv_bf16x3_lds_kernel(ConvP p) {
    constexpr int NP = TERMS == 1 ? 1 : 3;
    __shared__ u32x4 lds_a[2][4][NP][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int tile = blockIdx.x * 4 + wave;
    const bool live = tile < p.tiles;                      // (a workgroup's waves all stay for the barriers)
    if (!live) tile = p.tiles - 1;
    const int T = blockIdx.y;
    const int tiles_per_row = p.W / 64;
    const int n = tile / (p.H * tiles_per_row);
    const int rem = tile - n * (p.H * tiles_per_row);
    const int yrow = rem / tiles_per_row;
    const int x0 = (rem - yrow * tiles_per_row) * 64;
    const int j = lane & 31, kb = lane >> 5;
    const int S = p.C / 16;
    const int KT = S * 9;

    f32x16 acc[4][2];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;

    u32x4 Areg[2][4][NP], B[3][2][NP], stage[2][NP];
    const u32x4 zero = {0u, 0u, 0u, 0u};
    const size_t wbase = (size_t)T * S * 9 * 4 * 64 + (size_t)wave * 64 + lane;       // this wave's band
    auto gload = [&](const int slot_b, const int slot_s, const int it) __attribute__((always_inline)) {
        const int s = it / 9, tap = it - s * 9;
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const int yy = yrow + dy;
        const size_t wofs = wbase + (size_t)it * 4 * 64;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) stage[slot_s][pl] = p.ws[pl * p.w_plane_u4 + wofs];
        const size_t crow = ((size_t)n * (p.C / 8) + 2 * s + kb) * p.H;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int xx = x0 + 32 * nb + j + dx;
            const bool inb = yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
            const size_t o = (crow + (size_t)(inb ? yy : 0)) * p.W + (inb ? xx : 0);
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                const u32x4 v = p.xs[pl * p.x_plane_u4 + o];
                B[slot_b][nb][pl] = inb ? v : zero;
            }
        }
    };
    auto to_lds = [&](const int buf, const int slot_s) __attribute__((always_inline)) {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) lds_a[buf][wave][pl][lane] = stage[slot_s][pl];
    };
    auto from_lds = [&](const int buf, const int set) __attribute__((always_inline)) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) Areg[set][mb][pl] = lds_a[buf][mb][pl][lane];
    };
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};      // weight plane     (smallest terms first: x0 w2, x2 w0, x1 w1, x0 w1, x1 w0, x0 w0)
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};      // activation plane
    constexpr int T0 = TERMS == 1 ? 5 : 0;
    auto mfmas = [&](const int set, const int slot_b, const int t_lo, const int t_hi) __attribute__((always_inline)) {
#pragma unroll
        for (int t = t_lo; t < t_hi; ++t)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Areg[set][mb][NP == 1 ? 0 : PA[t]]),
                                                                          __builtin_bit_cast(bf16x8, B[slot_b][nb][NP == 1 ? 0 : PB[t]]),
                                                                          acc[mb][nb], 0, 0, 0);
    };
    // the workgroup barrier WITHOUT __syncthreads()' fence: that one also waits for every outstanding global load (vmcnt(0)), i.e. it
    // would drain the two-steps-ahead operand requests at every k-step; only this wave's LDS traffic has to have landed
    auto wg_barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    // prologue: steps 0 and 1 requested, step 0's weights through LDS into register set 0
    gload(0, 0, 0);
    if (KT > 1) gload(1, 1, 1);
    to_lds(0, 0);
    wg_barrier();
    from_lds(0, 0);
    // body `it` (unrolled by 6 so that every slot index is a compile-time constant: B ring of 3, stage / LDS / register sets of 2)
    auto step = [&](const int it, const int u) __attribute__((always_inline)) {
        if (it + 1 < KT) to_lds((u + 1) & 1, (u + 1) & 1);             // weights of step it + 1 (requested one body ago)
        if (it + 2 < KT) gload((u + 2) % 3, u & 1, it + 2);            // operands of step it + 2
        mfmas(u & 1, u % 3, T0, TERMS == 1 ? 6 : 3);
        wg_barrier();
        if (it + 1 < KT) from_lds((u + 1) & 1, (u + 1) & 1);
        if (TERMS != 1) mfmas(u & 1, u % 3, 3, 6);
    };
    for (int it = 0; it < KT; it += 6) {               // (KT = 9 C / 16 is a multiple of 6 whenever C % 32 == 0)
#pragma unroll
        for (int u = 0; u < 6; ++u) step(it + u, u);
    }
    if (!live) return;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = T * 128 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb;
            const float b = p.bias ? p.bias[co] : 0.0f;
            float* row = p.y + (((size_t)n * p.Cout + co) * p.H + yrow) * p.W + x0 + j;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) row[32 * nb] = acc[mb][nb][r] + b;
        }
}

// ---- v2: v1 was bound by the L2 -> L1 path, not by the matrix pipe (time proportional to the bytes requested: one plane 600 us,
// three planes 1500 us, against 553 us of MFMA time): with 16-byte operands the three input rows of a 16-channel group are 76 KB
// per CU, more than L1 holds, so each of the nine taps pulled its pixels out of L2 again (5.4 GB per launch).  Here a workgroup is
// FOUR ROWS x 64 pixels (wave w = row y0 + w) and stages the 6 x 66-pixel input patch of a channel group in LDS once per group,
// double-buffered (76 KB); the nine taps of the group then read their operands from LDS at shifted pixel addresses (16 bytes per
// pixel and lane: conflict-free).  Global traffic per wave and group: 12 patch loads + 27 weight loads instead of 81.
// ABL (timing ablations, results wrong by construction): 1 = MFMAs only inside the loop, 2 = no workgroup barrier, 3 = no global loads in the loop
template <int TERMS, int ABL = 0>
__global__ __launch_bounds__(256, 1) void conv_bf16x3_patch_kernel(ConvP p) {
    constexpr int NP = TERMS == 1 ? 1 : 3;
    __shared__ u32x4 lds_a[2][4][NP][64];
    __shared__ u32x4 lds_b[2][NP][6][2][66];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = blockIdx.y;
    const int per_row = p.W / 64;
    const int wg = blockIdx.x;
    const int n = wg / ((p.H / 4) * per_row);
    const int rem = wg - n * ((p.H / 4) * per_row);
    const int y0 = (rem / per_row) * 4;
    const int x0 = (rem - (rem / per_row) * per_row) * 64;
    const int yrow = y0 + wave;
    const int j = lane & 31, kb = lane >> 5;
    const int S = p.C / 16;

    f32x16 acc[4][2];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;

    u32x4 Areg[2][4][NP], Breg[2][2][NP], stage[4][NP], pstage[4][NP];
    const u32x4 zero = {0u, 0u, 0u, 0u};
    const size_t wbase = (size_t)T * S * 9 * 4 * 64 + (size_t)wave * 64 + lane;       // this wave's band
    auto wg_barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto a_gload = [&](const int slot, const int it) __attribute__((always_inline)) {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) stage[slot][pl] = p.ws[pl * p.w_plane_u4 + wbase + (size_t)it * 4 * 64];
    };
    auto a_to_lds = [&](const int buf, const int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) lds_a[buf][wave][pl][lane] = stage[slot][pl];
    };
    auto a_from_lds = [&](const int buf, const int set) __attribute__((always_inline)) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) Areg[set][mb][pl] = lds_a[buf][mb][pl][lane];
    };
    // the patch of channel group s: 6 rows x 2 channel octets x 66 pixels of 16 bytes per plane, four items per thread
    auto patch_gload = [&](const int s) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + 256 * i;
            const int r = q / 132, rem2 = q - r * 132, kbb = rem2 / 66, px = rem2 - kbb * 66;
            const int yy = y0 - 1 + r, xx = x0 - 1 + px;
            const bool inb = q < 792 && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
            const size_t o = (((size_t)n * (p.C / 8) + 2 * s + kbb) * p.H + (size_t)(inb ? yy : 0)) * p.W + (inb ? xx : 0);
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                const u32x4 v = p.xs[pl * p.x_plane_u4 + o];
                pstage[i][pl] = inb ? v : zero;
            }
        }
    };
    auto patch_to_lds = [&](const int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + 256 * i;
            if (q < 792) {
                const int r = q / 132, rem2 = q - r * 132, kbb = rem2 / 66, px = rem2 - kbb * 66;
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) lds_b[buf][pl][r][kbb][px] = pstage[i][pl];
            }
        }
    };
    auto b_from_lds = [&](const int buf, const int tap, const int set) __attribute__((always_inline)) {
        const int dy = tap / 3, dx = tap - (tap / 3) * 3;            // (0 .. 2: the patch starts one row / pixel before the tile)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) Breg[set][nb][pl] = lds_b[buf][pl][wave + dy][kb][32 * nb + j + dx];
    };
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};      // weight plane     (smallest terms first)
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};      // activation plane
    constexpr int T0 = TERMS == 1 ? 5 : 0;
    auto mfmas = [&](const int set, const int t_lo, const int t_hi) __attribute__((always_inline)) {
#pragma unroll
        for (int t = t_lo; t < t_hi; ++t)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Areg[set][mb][NP == 1 ? 0 : PA[t]]),
                                                                          __builtin_bit_cast(bf16x8, Breg[set][nb][NP == 1 ? 0 : PB[t]]),
                                                                          acc[mb][nb], 0, 0, 0);
    };
    const int KT = S * 9;
    // prologue: the weights run THREE k-steps ahead in a ring of four staging slots (one step ahead the L2 latency of the request
    // was exposed at the top of every step: `s_waitcnt vmcnt(0)` right behind the load)
    patch_gload(0);
    a_gload(0, 0);
    patch_to_lds(0);
    a_to_lds(0, 0);
    a_gload(1, 1);
    a_gload(2, 2 < KT ? 2 : KT - 1);
    wg_barrier();
    a_from_lds(0, 0);
    b_from_lds(0, 0, 0);
    // four channel groups (36 k-steps) per body: every register set / ring slot / LDS buffer index is a compile-time constant.
    // NO conditional inside the body -- past the end the requests are clamped to the last step / group (a few redundant loads
    // whose data is never used): a branch per step makes every step its own basic block and hipcc's wait-count pass then drains
    // the vector memory queue several times per step.  The first body is peeled for the same pass (loop-entry vs back-edge state).
    auto body = [&](const int s4) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 36; ++u) {
            const int g = u / 9, tap = u - g * 9;                  // group within the body, tap
            const int s = s4 + g, it = s * 9 + tap;
            // (scheduling fences between the phases: left alone, hipcc sinks every global load to just in front of its use --
            //  one step's latency fully exposed -- and interleaves the LDS reads with the MFMAs that wait for them)
            if (ABL != 1) a_to_lds((u + 1) & 1, (u + 1) & 3);        // weights of step it + 1 (requested two steps ago)
            if (ABL != 1 && ABL != 3) a_gload((u + 3) & 3, it + 3 < KT ? it + 3 : KT - 1);
            if (ABL != 1 && ABL != 3 && tap == 0) patch_gload(s + 1 < S ? s + 1 : S - 1);    // the next group's patch: requested here ...
            if (ABL != 1 && tap == 5) patch_to_lds((g + 1) & 1);     // ... parked five steps later, visible after the barriers of taps 5 .. 8
            __builtin_amdgcn_sched_barrier(0);
            mfmas(u & 1, T0, TERMS == 1 ? 6 : 3);
            __builtin_amdgcn_sched_barrier(0);
            if (ABL != 1 && ABL != 2) wg_barrier();
            if (ABL != 1) {
                a_from_lds((u + 1) & 1, (u + 1) & 1);
                b_from_lds(tap == 8 ? (g + 1) & 1 : g & 1, tap == 8 ? 0 : tap + 1, (u + 1) & 1);
            } else {
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) for (int pl = 0; pl < NP; ++pl) asm volatile("" : "+v"(Areg[(u + 1) & 1][mb][pl]));
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) for (int pl = 0; pl < NP; ++pl) asm volatile("" : "+v"(Breg[(u + 1) & 1][nb][pl]));
            }
            __builtin_amdgcn_sched_barrier(0);
            if (TERMS != 1) mfmas(u & 1, 3, 6);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    body(0);
    for (int s4 = 4; s4 < S; s4 += 4) body(s4);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = T * 128 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb;
            const float b = p.bias ? p.bias[co] : 0.0f;
            float* row = p.y + (((size_t)n * p.Cout + co) * p.H + yrow) * p.W + x0 + j;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) row[32 * nb] = acc[mb][nb][r] + b;
        }
}

}  // namespace

extern "C" int mcq_probe_split_bf16x3(const float* x, void* out, int N, int C, int HW, void* stream) {
    if (!x || !out || N <= 0 || C <= 0 || (C & 7) || HW <= 0) return -1;
    const size_t total = (size_t)N * (C / 8) * HW;
    hipLaunchKernelGGL(split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (u32x4*)out, N, C, HW, total);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int mcq_probe_conv3x3_bf16x3(const void* xs, const void* ws, const float* bias, float* y, int N, int C, int H, int W, int Cout,
                                        int terms, void* stream) {
    if (!xs || !ws || !y || N <= 0 || C <= 0 || (C % 64) || H <= 0 || W <= 0 || (W % 64) || Cout <= 0 || (Cout % 128)) return -1;
    ConvP p;
    p.xs = (const u32x4*)xs; p.ws = (const u32x4*)ws; p.bias = bias; p.y = y;
    p.N = N; p.C = C; p.H = H; p.W = W; p.Cout = Cout; p.tiles = N * H * (W / 64);
    p.x_plane_u4 = (size_t)N * (C / 8) * H * W;
    p.w_plane_u4 = (size_t)(Cout / 128) * (C / 16) * 9 * 4 * 64;
    p.terms = terms;
    const dim3 grid((unsigned)((p.tiles + 3) / 4), (unsigned)(Cout / 128));
    // terms: 6 / 1 = v2 (input patch + weights through LDS; needs H % 4 == 0, C % 32 == 0), 61 / 11 = v1 (weights through LDS),
    //        60 / 10 = v0 (every operand straight from global memory)
    if ((terms == 6 || terms == 1) && H % 4 == 0 && C % 64 == 0) {
        const dim3 g2((unsigned)(N * (H / 4) * (W / 64)), (unsigned)(Cout / 128));
        if (terms == 1) hipLaunchKernelGGL(conv_bf16x3_patch_kernel<1>, g2, dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL(conv_bf16x3_patch_kernel<6>, g2, dim3(256), 0, (hipStream_t)stream, p);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    if (terms >= 601 && terms <= 603 && H % 4 == 0 && C % 64 == 0) {          // timing ablations of v2
        const dim3 g2((unsigned)(N * (H / 4) * (W / 64)), (unsigned)(Cout / 128));
        if (terms == 601) hipLaunchKernelGGL((conv_bf16x3_patch_kernel<6, 1>), g2, dim3(256), 0, (hipStream_t)stream, p);
        else if (terms == 602) hipLaunchKernelGGL((conv_bf16x3_patch_kernel<6, 2>), g2, dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((conv_bf16x3_patch_kernel<6, 3>), g2, dim3(256), 0, (hipStream_t)stream, p);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    if (terms == 10) hipLaunchKernelGGL(conv_bf16x3_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else if (terms == 60) hipLaunchKernelGGL(conv_bf16x3_kernel<6>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else if (terms == 1 || terms == 11) hipLaunchKernelGGL(conv_bf16x3_lds_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(conv_bf16x3_lds_kernel<6>, grid, dim3(256), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
