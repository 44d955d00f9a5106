// Weight gradients of 3x3 stride-1 / 1x1 convolutions over FEW pixels (N H W <= 512: the 4x4 and 8x8 maps of a training step on
// 256x256 crops), on v_mfma_f32_16x16x4_f32 (included by wgrad_rows.hip).
//
//   dW[co][ci][dy][dx] = sum_{n, y, x} dY[n][co][y][x] * X[n][ci][y + dy - 1][x + dx - 1]          db[co] = sum dY[n][co][y][x]
//
// The strip walk of wgrad_rows.hip owns 32 x 32 tiles with 144 accumulator registers and cuts the pixel rows into ranges whose
// partial sums a second pass adds up; on these maps it is all fixed cost: sixteen 8 x 128 x 8 x 8 problems in one launch took 92 us
// for 15 us of MFMA work, and the 4x4 level ran a lane-per-weight VALU kernel out of LDS (twelve problems: 69 us for 3 us of
// work).  Here a workgroup owns a 16 co x 16 ci tile with ALL its taps (TAPS x 4 accumulator registers), the contraction runs over
// the flattened pixels, four per instruction:
//     A[i = l & 15][k = l >> 4] = dY[pixel 4 s + k][co0 + i]        B[k = l >> 4][j = l & 15] = X[pixel 4 s + k shifted by the tap][ci0 + j]
//     D[4 (l >> 4) + r][l & 15] = dW[co0 + 4 (l >> 4) + r][ci0 + (l & 15)][tap]
// its four waves take every fourth step and meet in LDS in wave order (deterministic, no second pass, no workspace).  The 16
// channels of dY and X the tile contracts are staged in LDS pixel-major first (64 KB at 512 pixels); out-of-image taps read 0; the
// bias gradient is the sum of a lane's own A operands.  One launch carries up to MCQ_WGRAD_MAX_GROUP convolutions (grid.z).
// What the per-workgroup timeline (s_memrealtime stamps in a scratch build, round 3) made of it, sixteen 8 x 128 x 8 x 8 problems =
// 1024 workgroups: operands straight from global memory (a lane's channel is a plane of its own: 64 cache lines per load
// instruction) 130 us; staged in LDS but one pixel per thread and round trip 92; the nine tap validities as && / ?: chains (hipcc:
// ~400 instructions of exec-mask control flow per pixel quad) 103; as plain comparisons combined with & 56; 16-byte staging loads
// with the next image group requested ahead 48 us -- against 92 + 28 us for the strip walk and its reduce pass.  Twelve 4x4
// problems 69 -> 21 us, a single 1x1 weight gradient 17-22 -> 8 us.
#pragma once

namespace {

struct WgT16K {
    const float* x[ROWS_MAX_CONVS]; const float* dy[ROWS_MAX_CONVS]; float* dw[ROWS_MAX_CONVS]; float* dbias[ROWS_MAX_CONVS];
    int N, Cin, Cout, H, W;
    int hw_log2, w_log2;     // >= 0: H W / W are powers of two (shifts instead of divisions in the pixel walk)
};

constexpr int WGT16_MAX_PIXELS = 512;        // N H W the kernel takes (at 2048 -- eight 16x16 images -- it only matches the strip walk: 167 vs 171 us for sixteen problems)
constexpr int WGT16_STAGE_PIXELS = 256;      // pixels (whole images) staged in LDS at a time: 32 KB per workgroup, four workgroups per CU
inline bool wgt16_shape(int N, int Cin, int H, int W, int Cout) {
    return N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && (long long)N * H * W <= WGT16_MAX_PIXELS && (H * W) % 4 == 0 && H * W <= WGT16_STAGE_PIXELS && Cin % 16 == 0 && Cout % 16 == 0 &&
           (uint64_t)N * (Cin > Cout ? Cin : Cout) * H * W * 4ull < 0x40000000ull;
}

template <int TAPS, bool SQ>
__global__ __launch_bounds__(256) void conv_wgrad_t16_kernel(WgT16K p) {
    // [pixel][16 channels] of dY (co0 ..) and X (ci0 ..): what the MFMA operands read -- lane (k, i) of step s reads channel i of
    // pixel 4 s + k, 64 consecutive floats per wave.  (The first version took the operands straight from global memory: there a
    // lane's channel is a plane of its own, every load instruction touched 64 cache lines, and sixteen 8x8 problems took 130 us.)
    __shared__ float stage[2][WGT16_STAGE_PIXELS][16];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int kq = lane >> 4, i = lane & 15;
    const int conv = blockIdx.z;
    const float* xp = p.x[0];
    const float* dyp = p.dy[0];
    float* dw = p.dw[0];
    float* dbias = p.dbias[0];
#pragma unroll
    for (int c = 1; c < ROWS_MAX_CONVS; ++c)
        if (c == conv) { xp = p.x[c]; dyp = p.dy[c]; dw = p.dw[c]; dbias = p.dbias[c]; }
    const int ci0 = (int)blockIdx.x * 16, co0 = (int)blockIdx.y * 16;
    const int HW = p.H * p.W, K = p.N * HW;
    int toff[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) toff[t] = TAPS == 9 ? (t / 3 - 1) * p.W + (t % 3 - 1) : 0;
    f32x4v acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) acc[t] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
    float bsum = 0.0f;
    // whole images at a time, as many as fit the staging area (64 KB for all 512 pixels at once allowed ONE workgroup per CU: sixteen
    // 8x8 problems = 1024 workgroups ran as four rounds, 90 us)
    const int NC = WGT16_STAGE_PIXELS / HW;
    // staging: thread (c = t & 15, g = t >> 4) moves channel c of pixel quads g, g + 16, g + 32, g + 48 of the image group -- one
    // 16-byte load per quad and tensor (a wave's load covers whole 64-byte runs of 16 channel planes), all eight in flight at once,
    // and the NEXT group's are requested before this group is multiplied.
    const int sc = threadIdx.x & 15, sg = threadIdx.x >> 4;
    f32x4v ld[4], lx[4];
    auto request = [&](const int n0, const int nc) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int P = 4 * (sg + 16 * u);                     // first pixel of the quad inside the group (H W % 4 == 0: one image)
            const bool live = P < nc * HW;
            const int Pc = live ? P : 0;
            const int n = Pc / HW, rem = Pc - n * HW;
            const f32x4v z = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
            ld[u] = live ? *reinterpret_cast<const f32x4v*>(dyp + ((size_t)(n0 + n) * p.Cout + co0 + sc) * HW + rem) : z;
            lx[u] = live ? *reinterpret_cast<const f32x4v*>(xp + ((size_t)(n0 + n) * p.Cin + ci0 + sc) * HW + rem) : z;
        }
    };
    auto deposit = [&](const int nc) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int P = 4 * (sg + 16 * u);
            if (P < nc * HW) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    stage[0][P + e][sc] = ld[u][e];
                    stage[1][P + e][sc] = SQ ? lx[u][e] * lx[u][e] : lx[u][e];
                }
            }
        }
    };
    request(0, p.N < NC ? p.N : NC);
    for (int n0 = 0; n0 < p.N; n0 += NC) {
        const int nc = p.N - n0 < NC ? p.N - n0 : NC;
        if (n0 > 0) __syncthreads();                             // (everybody has finished reading the previous group)
        deposit(nc);
        __syncthreads();
        if (n0 + NC < p.N) request(n0 + NC, p.N - (n0 + NC) < NC ? p.N - (n0 + NC) : NC);
        // The contraction order is free: quads of pixel POSITIONS outermost, images innermost -- the tap geometry of a lane (which
        // of its nine neighbours exist, and where) then depends on the position only and is worked out once per image group; inside,
        // a step costs one select per tap beside its ten LDS reads and nine MFMAs.  Wave w takes images w, w + 4, ... of the group.
        const int quads = HW >> 2;                               // (launcher: H W % 4 == 0)
        // the group's (quad, image) steps over the four waves: by image when there are four or more, by quad when the group is a
        // single image (a 16x16 map fills the staging area alone), two by two in between
        const int nstep = nc >= 3 ? 4 : nc, nstart = nc >= 3 ? wave : wave % nc;
        const int qstep = 4 / nstep, qstart = nc >= 3 ? 0 : wave / nc;
        const int rem0 = 4 * qstart + kq;
        int x = rem0 % p.W, y = rem0 / p.W;                      // this lane's position in its first quad; + 4 qstep pixels per quad, no divisions
        for (int q = qstart; q < quads; q += qstep) {
            const int rem = 4 * q + kq;
            // (plain comparisons combined with &: written with && / ?: chains hipcc turned the nine validities into ~400
            //  instructions of exec-mask control flow per quad, 19 us of a workgroup's 24)
            const bool r0 = y > 0, r2 = y < p.H - 1, c0 = x > 0, c2 = x < p.W - 1;
            bool ok[TAPS];
            int off[TAPS];
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                ok[t] = true;
                if (TAPS == 9) {
                    const bool rv = t / 3 == 0 ? r0 : t / 3 == 2 ? r2 : true;
                    const bool cv = t % 3 == 0 ? c0 : t % 3 == 2 ? c2 : true;
                    ok[t] = rv & cv;
                }
                off[t] = (rem + (ok[t] ? toff[t] : 0)) * 16 + i;
            }
            x += 4 * qstep;
            while (x >= p.W) { x -= p.W; ++y; }
            for (int n = nstart; n < nc; n += nstep) {
                const int base = n * HW * 16;
                const float a = stage[0][0][base + rem * 16 + i];
                bsum += a;
                float b[TAPS];
#pragma unroll
                for (int t = 0; t < TAPS; ++t) {
                    const float v = stage[1][0][base + off[t]];
                    b[t] = ok[t] ? v : 0.0f;
                }
#pragma unroll
                for (int t = 0; t < TAPS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[t], acc[t], 0, 0, 0);
            }
        }
    }
    // bias: lanes (kq, i) of all four kq hold parts of column co0 + i
    bsum += __shfl_xor(bsum, 16);
    bsum += __shfl_xor(bsum, 32);
    __syncthreads();                                             // (the staging area becomes the meeting place of the four waves)
    f32x4v* part = reinterpret_cast<f32x4v*>(&stage[0][0][0]);   // [3][TAPS][64]
    float* part_b = reinterpret_cast<float*>(part + 3 * TAPS * 64);   // [3][64], behind the tiles (28.4 KB of the 32 in all)
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < TAPS; ++t) part[((wave - 1) * TAPS + t) * 64 + lane] = acc[t];
        part_b[(wave - 1) * 64 + lane] = bsum;
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        const f32x4v v = ((acc[t] + part[(0 * TAPS + t) * 64 + lane]) + part[(1 * TAPS + t) * 64 + lane]) + part[(2 * TAPS + t) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) dw[((size_t)(co0 + 4 * kq + r) * p.Cin + ci0 + i) * TAPS + t] = v[r];
    }
    if (dbias && blockIdx.x == 0 && kq == 0) dbias[co0 + i] = ((bsum + part_b[lane]) + part_b[64 + lane]) + part_b[128 + lane];
}

inline int log2_or_minus1(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

// launch for `nconv` convolutions of one shape; `square_x`: the contraction runs over x^2 (GDN's gamma, 1x1 only)
inline void wgt16_launch(const float* const* x, const float* const* dy, float* const* dw, float* const* dbias, int nconv, int N, int Cin,
                         int H, int W, int Cout, int taps, bool square_x, hipStream_t s) {
    WgT16K t;
    for (int c = 0; c < ROWS_MAX_CONVS; ++c) {
        const int k = c < nconv ? c : 0;
        t.x[c] = x[k]; t.dy[c] = dy[k]; t.dw[c] = dw[k]; t.dbias[c] = dbias ? dbias[k] : nullptr;
    }
    t.N = N; t.Cin = Cin; t.Cout = Cout; t.H = H; t.W = W;
    t.hw_log2 = log2_or_minus1(H * W); t.w_log2 = log2_or_minus1(W);
    const dim3 grid((unsigned)(Cin / 16), (unsigned)(Cout / 16), (unsigned)nconv);
    if (taps == 9) hipLaunchKernelGGL((conv_wgrad_t16_kernel<9, false>), grid, dim3(256), 0, s, t);
    else if (square_x) hipLaunchKernelGGL((conv_wgrad_t16_kernel<1, true>), grid, dim3(256), 0, s, t);
    else hipLaunchKernelGGL((conv_wgrad_t16_kernel<1, false>), grid, dim3(256), 0, s, t);
}

}  // namespace
