// Backward of the soft assignment (BASELINE config #5), the two contractions over the [rows, k] factor d dist as exact-fp32 MFMA
// GEMMs (v_mfma_f32_32x32x2_f32) -- reference: the autograd of mcquic/modules/quantizer.py:181-183 (`_distance`) under :262-274:
//
//   dx[v][j]     = 2 x[v][j] rowsum[v]  - 2 sum_c ddist[v][c] C[g][c][j]
//   dC[g][c][j]  = 2 C[g][c][j] colsum_c - 2 sum_v ddist[v][c] x[v][j] + sum_{v: index_v = c} hot_v dDeq[v][j]
//
// ddist is 134 MB at the first level of the training geometry (8 x 2 x 16 x 16 vectors, k = 8192) and each contraction is
// 4.3 GFLOP: 27 us of MFMA time, 34 us of HBM time.  The lane-per-channel VALU forms in vq_train.hip (kept for geometries these
// kernels do not take) needed 400 / 352 us: one FMA per lane and scalar load.
#include "mcq_common.h"
#include "vq_bwd_mfma.h"

namespace {

__device__ __forceinline__ f32x4v load4_s(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

// ---- d codebook: D[c][j] over v.  A workgroup owns 32 codewords of one group; its eight waves are eight slices of the vectors
// and meet in LDS in wave order (deterministic).  (Four waves with an 8-step ring: 80 us at k = 8192 against 27 us of MFMA time --
// every ddist line is a first touch from HBM, and 8 steps x 128 MFMA cycles is less than that latency.)  A operand: lane (hi, i) = ddist[v = 2 s + hi][c0 + i] -- 128 contiguous bytes
// per half-wave; B operand: lane (hi, j) = x[v][g d + j] from the channel-major copy, two 32-channel blocks.  The column sum of
// ddist rides along on the VALU (one add per step); the straight-through term -- hot_v dDeq_v into the row of v's sampled
// codeword -- is sparse (one row per vector): every wave scans the indices of its slice, 64 at a time, and adds the hits in
// vector order.
constexpr int DC_PF = 16, DC_WAVES = 8;
__global__ __launch_bounds__(64 * DC_WAVES) void vq_dc_mfma_kernel(VqBwdK p) {
    __shared__ float part[DC_WAVES - 1][2][16][64];
    __shared__ float part_cs[DC_WAVES - 1][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int hi = lane >> 5, i = lane & 31;
    const int g = blockIdx.y, c0 = blockIdx.x * 32;
    const int C = p.m * p.d, V = p.N * p.hw;
    const int per = V / DC_WAVES;                                    // V % (2 DC_WAVES) == 0 (launcher): whole vector pairs per slice
    const int v0 = wave * per;
    const __amdgpu_buffer_rsrc_t dr = mcq_make_rsrc(p.ddist, (unsigned)((size_t)p.rows * p.k * 4u));
    const __amdgpu_buffer_rsrc_t xr = mcq_make_rsrc(p.xt, (unsigned)((size_t)V * C * 4u));
    const unsigned avo = ((unsigned)hi * (unsigned)p.k + (unsigned)(c0 + i)) * 4u;
    unsigned bvo[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) bvo[nb] = 32 * nb + i < p.d ? ((unsigned)hi * (unsigned)C + (unsigned)(g * p.d + 32 * nb + i)) * 4u : MCQ_OOB;
    // load cursor: vector pair (n, pp), pp even; row of ddist = (n m + g) hw + pp
    int ln = v0 / p.hw, lp = v0 - ln * p.hw;
    float A[DC_PF], B[DC_PF][2];
    auto issue = [&](const int st) __attribute__((always_inline)) {
        const unsigned row = (unsigned)((ln * p.m + g) * p.hw + lp), vec = (unsigned)(ln * p.hw + lp);
        A[st] = mcq_buffer_load_s(dr, avo, row * (unsigned)p.k * 4u);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) B[st][nb] = mcq_buffer_load_s(xr, bvo[nb], vec * (unsigned)C * 4u);
        lp += 2;
        if (lp >= p.hw) { lp = 0; ++ln; }                            // (hw even)
    };
    f32x16 acc[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.0f;
    float cs = 0.0f;
#pragma unroll
    for (int st = 0; st < DC_PF; ++st) issue(st);
    const int steps = per >> 1;
    for (int s = 0; s < steps; s += DC_PF) {
#pragma unroll
        for (int st = 0; st < DC_PF; ++st) {
            const float a = s + st < steps ? A[st] : 0.0f;          // (the ring runs past the slice: those are the next wave's vectors)
            cs += a;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, B[st][nb], acc[nb], 0, 0, 0);
            issue(st);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = -2.0f * acc[nb][r];
    // straight-through rows of this slice, in vector order
    for (int q = 0; q < per; q += 64) {
        const int v = v0 + q + lane;
        int rel = -1;
        unsigned row = 0;
        if (q + lane < per) {
            const int n = v / p.hw, pp = v - n * p.hw;
            row = (unsigned)((n * p.m + g) * p.hw + pp);
            rel = (int)p.index[row] - c0;
        }
        unsigned long long hits = __ballot(rel >= 0 && rel < 32);
        while (hits) {
            const int src = __builtin_ctzll(hits);
            hits &= hits - 1;
            const int rsel = __builtin_amdgcn_readlane(rel, src);
            const unsigned rrow = (unsigned)__builtin_amdgcn_readlane((int)row, src);
            const unsigned vv = (unsigned)(v0 + q + src);
            const float hv = p.hot[rrow];
            const int reg = (rsel & 3) + 4 * (rsel >> 3);
            const bool mine = hi == ((rsel >> 2) & 1);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const float t = 32 * nb + i < p.d ? hv * p.dqt[(size_t)vv * C + (size_t)(g * p.d + 32 * nb + i)] : 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nb][r] = (mine && r == reg) ? acc[nb][r] + t : acc[nb][r];
            }
        }
    }
    cs += __shfl_xor(cs, 32);                                        // even + odd vectors: lane i (both halves) = column c0 + i
    if (wave > 0) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) part[wave - 1][nb][r][lane] = acc[nb][r];
        part_cs[wave - 1][lane] = cs;
    }
    __syncthreads();
    if (wave != 0) return;
    float col_i = cs;
#pragma unroll
    for (int w = 0; w < DC_WAVES - 1; ++w) col_i += part_cs[w][lane];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int cr = mcq_drow(r, hi);
        const float col = __shfl(col_i, cr);
        const int c = c0 + cr;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int j = 32 * nb + i;
            if (j < p.d && c < p.k) {
                float t = acc[nb][r];
#pragma unroll
                for (int w = 0; w < DC_WAVES - 1; ++w) t += part[w][nb][r][lane];
                const size_t ci = ((size_t)g * p.k + c) * p.d + j;
                p.dcb[ci] = 2.0f * p.cb[ci] * col + t;
            }
        }
    }
}

// ---- dx: D[j][v] over the codewords.  A workgroup owns 32 consecutive vectors of one (image, group); its sixteen waves are
// sixteen slices of the codewords (the tile count is rows / 32 = 128 at the first level: the split has to come from inside the
// workgroup -- and, below 256 tiles, from a second workgroup per tile, see the launcher) and meet in LDS as a fixed binary tree.  ddist is contiguous along the codewords and a lane owns a VECTOR, so a
// lane reads 16 bytes = four codewords of its row per access and the k-steps take the codewords in the order the lanes hold them:
// step (q, t) contracts codeword 8 q + 4 hi + t on half `hi`.  A operand: lane (hi, i) = C[g][8 q + 4 hi + t][32 jb + i].
constexpr int DX_PFQ = 4;
__global__ __launch_bounds__(1024) void vq_dx_mfma_kernel(VqBwdK p) {
    __shared__ float part[8][2][16][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int hi = lane >> 5, i = lane & 31;
    const int row0 = blockIdx.x * 32;                                // hw % 32 == 0: the 32 rows share (n, g)
    const int pix0 = row0 % p.hw, ng = row0 / p.hw, g = ng % p.m;
    const int kslice = p.k / (16 * (int)gridDim.y), kk0 = ((int)blockIdx.y * 16 + wave) * kslice;     // (launcher: kslice % 8 == 0)
    const __amdgpu_buffer_rsrc_t dr = mcq_make_rsrc(p.ddist, (unsigned)((size_t)p.rows * p.k * 4u));
    const __amdgpu_buffer_rsrc_t cr = mcq_make_rsrc(p.cb, (unsigned)((size_t)p.m * p.k * p.d * 4u));
    const unsigned bvo = ((unsigned)i * (unsigned)p.k + 4u * (unsigned)hi) * 4u;
    unsigned avo[2][4];
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int t = 0; t < 4; ++t)
            avo[jb][t] = 32 * jb + i < p.d ? ((unsigned)(4 * hi + t) * (unsigned)p.d + (unsigned)(32 * jb + i)) * 4u : MCQ_OOB;
    unsigned bso = ((unsigned)row0 * (unsigned)p.k + (unsigned)kk0) * 4u;
    unsigned aso = (unsigned)((g * p.k + kk0) * p.d) * 4u;
    f32x4v B4[DX_PFQ];
    float A[DX_PFQ][4][2];
    auto issue = [&](const int st) __attribute__((always_inline)) {
        B4[st] = load4_s(dr, bvo, bso);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) A[st][t][jb] = mcq_buffer_load_s(cr, avo[jb][t], aso);
        bso += 32u;
        aso += 8u * (unsigned)p.d * 4u;
    };
    f32x16 acc[2];
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[jb][r] = 0.0f;
#pragma unroll
    for (int st = 0; st < DX_PFQ; ++st) issue(st);
    const int chunks = kslice >> 3;
    for (int q = 0; q < chunks; q += DX_PFQ) {
#pragma unroll
        for (int st = 0; st < DX_PFQ; ++st) {
            const bool live = q + st < chunks;                      // (the ring runs past the slice: the next wave's codewords)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float b = live ? B4[st][t] : 0.0f;
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) acc[jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[st][t][jb], b, acc[jb], 0, 0, 0);
            }
            issue(st);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // sixteen slices -> one, as a fixed tree: (w, w + 8), (w, w + 4), (w, w + 2), (w, w + 1)
#pragma unroll
    for (int half = 8; half >= 1; half >>= 1) {
        if (wave >= half && wave < 2 * half) {
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int r = 0; r < 16; ++r) part[wave - half][jb][r][lane] = acc[jb][r];
        }
        __syncthreads();
        if (wave < half) {
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[jb][r] += part[wave][jb][r][lane];
        }
        __syncthreads();
    }
    if (wave != 0) return;
    const float rs = p.rowsum[row0 + i];
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = 32 * jb + mcq_drow(r, hi);
            if (j < p.d) {
                const size_t xi = ((size_t)ng * p.d + j) * p.hw + pix0 + i;
                if (gridDim.y == 1) p.dx[xi] = 2.0f * p.x[xi] * rs - 2.0f * acc[jb][r];
                // two workgroups per tile (halves of the codewords): each adds its part onto the zeroed output -- with exactly two
                // addends the sum does not depend on who comes first (0 + a is a, a + b is b + a)
                else unsafeAtomicAdd(p.dx + xi, blockIdx.y == 0 ? 2.0f * p.x[xi] * rs - 2.0f * acc[jb][r] : -2.0f * acc[jb][r]);
            }
        }
}

// (a kernel, not hipMemsetAsync: a memset NODE in a replayed hipGraph was seen to run out of order with the kernel nodes around it
//  under ROCm 7.2's packet-captured graph launches -- tools/probes/replay_determinism.py, docs/experiments.md section 9.9)
__global__ __launch_bounds__(256) void vq_zero_kernel(float4* __restrict__ out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) out[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

}  // namespace

bool mcq_vq_dc_mfma_ok(const VqBwdK& p) {
    const long long V = (long long)p.N * p.hw;
    return p.d <= 64 && p.k % 32 == 0 && p.hw % 2 == 0 && V % (2 * DC_WAVES) == 0 && (unsigned long long)p.rows * p.k * 4ull < 0x80000000ull &&
           (unsigned long long)V * p.m * p.d * 4ull < 0x80000000ull;
}
bool mcq_vq_dx_mfma_ok(const VqBwdK& p) {
    return p.d <= 64 && p.hw % 32 == 0 && p.k % 128 == 0 && (unsigned long long)p.rows * p.k * 4ull < 0x80000000ull &&
           (unsigned long long)p.m * p.k * p.d * 4ull < 0x80000000ull;
}
void mcq_vq_dc_mfma_launch(const VqBwdK& p, void* stream) {
    hipLaunchKernelGGL(vq_dc_mfma_kernel, dim3((unsigned)(p.k / 32), (unsigned)p.m), dim3(64 * DC_WAVES), 0, (hipStream_t)stream, p);
}
void mcq_vq_dx_mfma_launch(const VqBwdK& p, void* stream) {
    // fewer tiles than CUs (rows / 32 = 128 at the first training level): the codewords are halved over two workgroups per tile
    const unsigned zs = (p.rows / 32 < 256 && p.k % 256 == 0) ? 2u : 1u;
    if (zs == 2) {                                       // rows % 32 == 0: a whole number of float4
        const size_t n4 = (size_t)p.rows * p.d / 4;
        hipLaunchKernelGGL(vq_zero_kernel, dim3((unsigned)((n4 + 255) / 256 < 1024 ? (n4 + 255) / 256 : 1024)), dim3(256), 0, (hipStream_t)stream,
                           (float4*)p.dx, n4);
    }
    hipLaunchKernelGGL(vq_dx_mfma_kernel, dim3((unsigned)(p.rows / 32), zs), dim3(1024), 0, (hipStream_t)stream, p);
}
