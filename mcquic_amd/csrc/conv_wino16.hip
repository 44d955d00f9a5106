// F(2x2, 3x3) on v_mfma_f32_16x16x4_f32: the TWO-WAVES-PER-SIMD form of the opt-in Winograd mode (round 3).
//
// The first 2-D instance (conv_mfma.hip, TAPS == 16) gives a wave one 32-row band x 32 tiles x 16 transform positions = 256
// accumulator registers: the whole AGPR half of a one-wave-per-SIMD register file.  Ablations put 22 % of its
// time OUTSIDE the k-loop -- per 28 us tile ~7.7 us of prologue (first operands from HBM) and epilogue (residual loads, 256
// accumulator reads) that a lone wave per SIMD exposes in full and cannot prefetch around (vmcnt retires in order).  The direct
// kernel hides exactly that with a second resident wave.  Here the same arithmetic runs on the 16 x 16 x 4 instruction -- the same
// FLOP rate, half the tile: a wave owns 32 output channels (two 16-row halves) x 16 tiles (2 x 2 pixels each) x 16 positions =
// 128 accumulator registers, compiler-managed, two waves per SIMD, so that one wave's prologue / epilogue runs beside the other's
// MFMAs.  Per group of FOUR input channels (the instruction's k) and wave: 32 MFMAs of 32 cycles = the 1 024 matrix cycles of a
// channel pair in the 32 x 32 x 2 form, for the same 64 (tile, channel) patches -- so the transform work per matrix cycle is the
// same; the price is weight operands: 32 loads per group instead of 16.
//
// Lane roles (v_mfma_f32_16x16x4_f32): lane l = 16 kq + t feeds A[row t][k kq] (weights of output channel 16 h + t, input channel
// 4 g + kq) and B[k kq][col t] (transformed input of tile t, channel 4 g + kq) and owns D[4 kq + r][t], r = 0 .. 3.
// The four waves of a workgroup are the four 32-row bands of 128 output channels over the SAME 16 tiles and share the input
// transform through LDS exactly like the first instance: wave w loads row w of every lane's 4 x 4 patch, transforms it along x,
// parks 4 floats; one workgroup barrier per channel group; every wave reads the four rows back and transforms along y.
// Weights: [band][group][position][half][64 lanes] floats, U = G g G^T in float64 rounded once (mcq_pack_conv_weight_winograd16_f32).
#include <type_traits>

#include "mcq_common.h"
#include "../../include/mcquic_hip.h"
#include "conv_wino16.h"

namespace {

typedef float f32x4a __attribute__((ext_vector_type(4)));

constexpr unsigned W16_RUNTIME = 0xffffffffu;

template <unsigned EF>
__global__ __launch_bounds__(256, 2) void conv_wino16_kernel(W16K p) {
    __shared__ f32x4v tl[2][4][64];                          // [buffer][patch row][lane]: x-transformed rows of the group in flight
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const float* P_x = p.x; const float* P_wp = p.wp; const float* P_bias = p.bias; float* P_y = p.y; float* P_y2 = p.y2;
    const float* P_res = p.res;
    if (p.nprob > 1) {
#pragma unroll
        for (int c = 1; c < W16_MAX_MULTI; ++c)
            if ((int)blockIdx.z == c) {
                const W16Ptrs& a = p.alt[c - 1];
                P_x = a.x; P_wp = a.wp; P_bias = a.bias; P_y = a.y; P_y2 = a.y2; P_res = a.res;
            }
    }
    // XCD-aware block order (see conv_mfma.hip): XCD k walks the contiguous k-th eighth of the tile blocks
    unsigned wg = blockIdx.x;
    {
        const unsigned nwg = gridDim.x, xcd = wg & 7u, slot = wg >> 3;
        const unsigned base = xcd * (nwg >> 3) + (xcd < (nwg & 7u) ? xcd : (nwg & 7u));
        wg = base + slot;
    }
    const int t = lane & 15, kq = lane >> 4;
    const int BW = 1 << p.bw_log2, BH = 16 >> p.bw_log2;      // a block is BH x BW tiles
    const int ty = t >> p.bw_log2, tx = t & (BW - 1);
    const int per_img = p.nby * p.nbx;
    const int n = (int)wg / per_img;
    const int rem = (int)wg - n * per_img;
    const int by = rem / p.nbx, bx = rem - by * p.nbx;
    const int y0 = 2 * (by * BH + ty), x0 = 2 * (bx * BW + tx);       // the tile's first output pixel
    const bool tile_ok = y0 < p.Ho && x0 < p.Wo;
    const int HW = p.H * p.W;
    const int co_base = ((int)blockIdx.y * 4 + wave) * 32;

    // ---- operand streams ----------------------------------------------------------------------------------------------------
    const char* xb = reinterpret_cast<const char*>(mcq_uniform_ptr(P_x + (size_t)n * p.Cin * HW));
    const unsigned plane_bytes = (unsigned)p.Cin * (unsigned)HW * 4u;
    unsigned vrow[4];                                         // my patch row (row `wave`), this lane's channel kq of the group
    {
        const int yi = y0 + wave - 1;
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            const int xi = x0 - 1 + px;
            const bool inb = tile_ok && yi >= 0 && yi < p.H && xi >= 0 && xi < p.W;
            vrow[px] = inb ? (unsigned)(yi * p.W + xi + kq * HW) * 4u : MCQ_OOB;
        }
    }
    const unsigned group_bytes = 4u * (unsigned)HW * 4u;       // four channels further
    const float* wbu = P_wp + (size_t)(co_base >> 5) * p.G * (32 * 64);
    const __amdgpu_buffer_rsrc_t wr = mcq_make_rsrc(wbu, 0x7fffffffu);      // (the pack ends in a zero tail: the ring's over-read is in bounds)
    const unsigned wlane = (unsigned)lane * 16u;              // 16 bytes per lane: the weights of four consecutive MFMAs
    unsigned wso = 0;

    f32x4a acc[16][2];
#pragma unroll
    for (int pos = 0; pos < 16; ++pos)
#pragma unroll
        for (int h = 0; h < 2; ++h) acc[pos][h] = f32x4a{0.0f, 0.0f, 0.0f, 0.0f};

    constexpr int GA = 2;                                      // groups the row loads run ahead = groups per loop body
    f32x4v A4[8];
    float Bw[GA][4];
#pragma unroll
    for (int q = 0; q < 8; ++q) { A4[q] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(wr, (int)wlane, (int)wso, 0)); wso += 1024; }
    {
        const __amdgpu_buffer_rsrc_t r0 = mcq_make_rsrc(xb, plane_bytes);
#pragma unroll
        for (int g = 0; g < GA; ++g)
#pragma unroll
            for (int px = 0; px < 4; ++px) Bw[g][px] = mcq_buffer_load(r0, vrow[px] + (unsigned)g * group_bytes);
    }
    auto row_to_lds = [&](const int slot, const int buf) __attribute__((always_inline)) {
        const float d0 = Bw[slot][0], d1 = Bw[slot][1], d2 = Bw[slot][2], d3 = Bw[slot][3];
        tl[buf][wave][lane] = f32x4v{d0 - d2, d1 + d2, d2 - d1, d1 - d3};
    };
    // (measured and dropped: SHARING the y-transform as well -- wave w forms only row w of B^T d B, a second LDS exchange and a
    //  second barrier per group hand the sixteen values round: 12 fewer vector-ALU instructions per group, 220/250 "TF" against
    //  224/251 for this form on the 384x256 layer.  The adds were not what the loop waits for.)
    auto lds_rows = [&](const int buf, f32x4v (&q)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int py = 0; py < 4; ++py) q[py] = tl[buf][py][lane];
    };
    auto rows_to_v = [&](const f32x4v (&q)[4], float (&out)[16]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            out[j] = q[0][j] - q[2][j]; out[4 + j] = q[1][j] + q[2][j]; out[8 + j] = q[2][j] - q[1][j]; out[12 + j] = q[1][j] - q[3][j];
        }
    };
    // the workgroup barrier WITHOUT the fence of __syncthreads() (that one waits for every outstanding vector load: the rings)
    auto wg_barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    float VV[2][16];                                           // transformed inputs of the group running / of the next one
    f32x4v tq[4];
    row_to_lds(0, 0);
    wg_barrier();
    lds_rows(0, tq);
    rows_to_v(tq, VV[0]);
    unsigned soff = 0;
    auto body = [&]() __attribute__((always_inline)) {
        __amdgpu_buffer_rsrc_t rB[GA];
#pragma unroll
        for (int j = 0; j < GA; ++j) {
            const unsigned off = soff + (unsigned)(GA + j) * group_bytes;
            const int left = (int)plane_bytes - (int)off;
            rB[j] = mcq_make_rsrc(xb + off, (unsigned)(left > 0 ? left : 0));
        }
#pragma unroll
        for (int u = 0; u < GA * 32; ++u) {
            const int k = u / 32, st = u % 32;                 // group within the body, MFMA within the group
            const int pos = st >> 1, h = st & 1;
            // the next group's transform spread over this group's steps (GA is even: VV[k & 1] / VV[(k + 1) & 1] keep their roles)
            if (st == 2) row_to_lds((k + 1) % GA, (k + 1) & 1);
            if (st == 12) { wg_barrier(); lds_rows((k + 1) & 1, tq); }
            if (st == 22) rows_to_v(tq, VV[(k + 1) & 1]);
            acc[pos][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(A4[st >> 2][st & 3], VV[k & 1][pos], acc[pos][h], 0, 0, 0);
            if (st < 4) Bw[k][st] = mcq_buffer_load(rB[k], vrow[st]);            // my row of the group GA groups ahead
            if ((st & 3) == 3) {
                A4[st >> 2] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(wr, (int)wlane, (int)wso, 0));
                wso += 1024;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        soff += (unsigned)GA * group_bytes;
    };
    // (no early exit inside a body: the launcher only takes Cin % 16 == 0 -- whole bodies; the first body is peeled for hipcc's
    //  wait-count pass, whose loop-entry state otherwise makes every step wait as if its operands had just been requested)
    if (p.G > 0) body();
    for (int g = GA; g < p.G; g += GA) body();

    // ---- epilogue: A^T M A per (half, register) = output channel, then bias / residual / SiLU / twin / store ----------------------
    const unsigned HoWo = (unsigned)(p.Ho * p.Wo);
    const unsigned fl = EF == W16_RUNTIME ? p.flags : EF;
    const bool shuffle = (fl & MCQ_CONV_SHUFFLE2) != 0;
    const unsigned ochan = shuffle ? (unsigned)p.Cout / 4u : (unsigned)p.Cout;
    const unsigned oHW = shuffle ? 4u * HoWo : HoWo;
    const unsigned slab_bytes = ochan * oHW * 4u;
    const size_t slab = (size_t)n * ochan * oHW;
    const __amdgpu_buffer_rsrc_t yr = mcq_make_rsrc(mcq_uniform_ptr(P_y + slab), slab_bytes);
    const __amdgpu_buffer_rsrc_t y2r = mcq_make_rsrc(mcq_uniform_ptr((fl & MCQ_CONV_DUAL_SILU) ? P_y2 + slab : P_y + slab), slab_bytes);
    const __amdgpu_buffer_rsrc_t rr = mcq_make_rsrc(mcq_uniform_ptr((fl & MCQ_CONV_RESIDUAL) ? P_res + slab : P_y + slab), slab_bytes);
    const __amdgpu_buffer_rsrc_t br = mcq_make_rsrc(mcq_uniform_ptr(P_bias ? P_bias : P_wp), P_bias ? (unsigned)p.Cout * 4u : 0u);
    const bool v0 = tile_ok, v1x = tile_ok && x0 + 1 < p.Wo, v1y = tile_ok && y0 + 1 < p.Ho;
    if (!shuffle) {
        // lane's pixels (oy, ox); with an even width the two pixels of a row are one 8-byte access
        const bool wide = (p.Wo & 1) == 0;
        unsigned pv[2][2];
#pragma unroll
        for (int oy = 0; oy < 2; ++oy)
#pragma unroll
            for (int ox = 0; ox < 2; ++ox) {
                const bool ok = v0 && (oy == 0 || v1y) && (ox == 0 || v1x);
                pv[oy][ox] = ok ? ((unsigned)((y0 + oy) * p.Wo + x0 + ox) + 4u * (unsigned)kq * HoWo) * 4u : MCQ_OOB;
            }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned row0 = (unsigned)co_base + 16u * (unsigned)h;           // + 4 kq (in pv) + r
            float bias4[4];
            f32x2v res2[2][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bias4[r] = mcq_buffer_load_s(br, (unsigned)kq * 16u, (row0 + (unsigned)r) * 4u);
                if (fl & MCQ_CONV_RESIDUAL) {
#pragma unroll
                    for (int oy = 0; oy < 2; ++oy) {
                        if (wide) res2[oy][r] = mcq_buffer_load2_s(rr, pv[oy][0], (row0 + (unsigned)r) * HoWo * 4u);
                        else res2[oy][r] = f32x2v{mcq_buffer_load_s(rr, pv[oy][0], (row0 + (unsigned)r) * HoWo * 4u),
                                                  mcq_buffer_load_s(rr, pv[oy][1], (row0 + (unsigned)r) * HoWo * 4u)};
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float sx[4][2];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    sx[i][0] = (acc[4 * i][h][r] + acc[4 * i + 1][h][r]) + acc[4 * i + 2][h][r];
                    sx[i][1] = (acc[4 * i + 1][h][r] - acc[4 * i + 2][h][r]) - acc[4 * i + 3][h][r];
                }
                const unsigned so = (row0 + (unsigned)r) * HoWo * 4u;
#pragma unroll
                for (int oy = 0; oy < 2; ++oy) {
                    f32x2v y, tw = {0.0f, 0.0f};
#pragma unroll
                    for (int ox = 0; ox < 2; ++ox) {
                        y[ox] = (oy == 0 ? (sx[0][ox] + sx[1][ox]) + sx[2][ox] : (sx[1][ox] - sx[2][ox]) - sx[3][ox]) + bias4[r];
                        if (fl & MCQ_CONV_RESIDUAL) y[ox] = y[ox] + p.res_scale * res2[oy][r][ox];
                    }
                    if (fl & MCQ_CONV_SILU_OUT) y = f32x2v{mcq_silu(y[0]), mcq_silu(y[1])};
                    if (fl & MCQ_CONV_DUAL_SILU) tw = f32x2v{mcq_silu(y[0]), mcq_silu(y[1])};
                    if (wide) {
                        mcq_buffer_store2_s(y, yr, pv[oy][0], so);
                        if (fl & MCQ_CONV_DUAL_SILU) mcq_buffer_store2_s(tw, y2r, pv[oy][0], so);
                    } else {
                        mcq_buffer_store_s(y[0], yr, pv[oy][0], so);
                        mcq_buffer_store_s(y[1], yr, pv[oy][1], so);
                        if (fl & MCQ_CONV_DUAL_SILU) {
                            mcq_buffer_store_s(tw[0], y2r, pv[oy][0], so);
                            mcq_buffer_store_s(tw[1], y2r, pv[oy][1], so);
                        }
                    }
                }
            }
        }
    } else {
        // PixelShuffle(2) store: the four registers r of a lane group are the 2 x 2 sub-pixels of ONE shuffled channel
        // c = (co_base + 16 h + 4 kq) / 4; the lane's 2 x 2 output pixels become a 4 x 4 block of that channel's plane
        const unsigned W2 = 2u * (unsigned)p.Wo;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned c = ((unsigned)co_base + 16u * (unsigned)h + 4u * (unsigned)kq) >> 2;
            float outv[4][2][2];                               // [r][oy][ox]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float b = mcq_buffer_load_s(br, (unsigned)kq * 16u, ((unsigned)co_base + 16u * (unsigned)h + (unsigned)r) * 4u);
                float sx[4][2];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    sx[i][0] = (acc[4 * i][h][r] + acc[4 * i + 1][h][r]) + acc[4 * i + 2][h][r];
                    sx[i][1] = (acc[4 * i + 1][h][r] - acc[4 * i + 2][h][r]) - acc[4 * i + 3][h][r];
                }
#pragma unroll
                for (int ox = 0; ox < 2; ++ox) {
                    outv[r][0][ox] = ((sx[0][ox] + sx[1][ox]) + sx[2][ox]) + b;
                    outv[r][1][ox] = ((sx[1][ox] - sx[2][ox]) - sx[3][ox]) + b;
                }
            }
            // shuffled row 2 (y0 + oy) + ry holds, for ox = 0, 1 and rx = 0, 1, sub-pixel r = 2 ry + rx of conv pixel (y0 + oy, x0 + ox)
#pragma unroll
            for (int oy = 0; oy < 2; ++oy)
#pragma unroll
                for (int ry = 0; ry < 2; ++ry)
#pragma unroll
                    for (int ox = 0; ox < 2; ++ox) {
                        const bool ok = v0 && (oy == 0 || v1y) && (ox == 0 || v1x);
                        const unsigned off = ok ? ((unsigned)(2 * (y0 + oy) + ry) * W2 + (unsigned)(2 * (x0 + ox))) * 4u : MCQ_OOB;
                        mcq_buffer_store2_s(f32x2v{outv[2 * ry][oy][ox], outv[2 * ry + 1][oy][ox]}, yr, off, c * oHW * 4u);
                    }
        }
    }
}

// U = G g G^T (float64, rounded once) into [band][group][position][half][lane]
__global__ void pack_wino16_kernel(const float* __restrict__ w, int Cout, int Cin, int G, float* __restrict__ out, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    // [band][group][quad q = st / 4][lane][e = st % 4], st = 2 pos + h: one 16-byte lane load feeds four MFMAs
    const int e = (int)(i & 3);
    const int lane = (int)((i >> 2) & 63);
    const int st = (int)((i >> 8) & 7) * 4 + e;
    const int h = st & 1, pos = st >> 1;
    const size_t bg = i >> 11;
    const int g = (int)(bg % (size_t)G);
    const int band = (int)(bg / (size_t)G);
    const int co = band * 32 + h * 16 + (lane & 15), ci = 4 * g + (lane >> 4);
    float v = 0.0f;
    if (co < Cout && ci < Cin) {
        const double Gm[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
        const int pi = pos >> 2, pj = pos & 3;
        double u = 0.0;
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) u += Gm[pi][a] * (double)w[((size_t)co * Cin + ci) * 9 + 3 * a + b] * Gm[pj][b];
        v = (float)u;
    }
    out[i] = v;
}

inline size_t wino16_floats(int Cout, int Cin) {             // + one group of zeros: the weight ring runs a group ahead
    return ((size_t)((Cout + 31) / 32) * (size_t)((Cin + 3) / 4) + 1) * 2048;
}

template <unsigned EF>
void launch_one(const W16K& k, dim3 grid, hipStream_t s) { hipLaunchKernelGGL(conv_wino16_kernel<EF>, grid, dim3(256), 0, s, k); }

}  // namespace

extern "C" size_t mcq_packed_conv_winograd16_floats(int32_t Cout, int32_t Cin) {
    return Cout <= 0 || Cin <= 0 ? 0 : wino16_floats(Cout, Cin);
}

extern "C" int mcq_pack_conv_weight_winograd16_f32(const float* w, int32_t Cout, int32_t Cin, float* out, void* stream) {
    if (!w || !out || Cout <= 0 || Cin <= 0) return MCQ_EINVAL;
    const size_t total = wino16_floats(Cout, Cin);
    hipLaunchKernelGGL(pack_wino16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin,
                       (Cin + 3) / 4, out, total);
    return mcq_check_launch();
}

int mcq_wino16_launch(W16K& k, void* stream) {
    if (k.Cout % 128 != 0 || k.Cin % 16 != 0 || k.nprob < 1 || k.nprob > W16_MAX_MULTI) return MCQ_EINVAL;
    if ((uint64_t)(k.Cin + 32) * k.H * k.W * 4ull >= 0x80000000ull) return MCQ_ETOOLARGE;
    k.G = k.Cin / 4;
    // tile blocks: 16 tiles of 2 x 2 pixels shaped (16 >> b) rows x (1 << b) tiles, b by the fewest wasted lanes (wider wins ties)
    const int Wt = (k.Wo + 1) / 2, Ht = (k.Ho + 1) / 2;
    int best_log2 = 4; double best_util = -1.0;
    for (int lg = 4; lg >= 0; --lg) {
        const int bw = 1 << lg, bh = 16 >> lg;
        const double cover = (double)((Ht + bh - 1) / bh * bh) * (double)((Wt + bw - 1) / bw * bw);
        const double util = (double)Ht * Wt / cover;
        if (util > best_util + 1e-9) { best_util = util; best_log2 = lg; }
    }
    k.bw_log2 = best_log2;
    k.nbx = (Wt + (1 << best_log2) - 1) >> best_log2;
    k.nby = (Ht + (16 >> best_log2) - 1) / (16 >> best_log2);
    const long long blocks = (long long)k.N * k.nbx * k.nby;
    if (blocks > 0x7fffffffLL) return MCQ_ETOOLARGE;
    const unsigned ochan = (k.flags & MCQ_CONV_SHUFFLE2) ? (unsigned)k.Cout / 4u : (unsigned)k.Cout;
    const uint64_t oHW = (uint64_t)k.Ho * k.Wo * ((k.flags & MCQ_CONV_SHUFFLE2) ? 4u : 1u);
    if ((uint64_t)ochan * oHW * 4ull >= 0x80000000ull) return MCQ_ETOOLARGE;
    const dim3 grid((unsigned)blocks, (unsigned)(k.Cout / 128), (unsigned)k.nprob);
    hipStream_t s = (hipStream_t)stream;
    const unsigned ef = k.flags & ~(unsigned)(MCQ_CONV_WINOGRAD2D16);
    if (ef == 0u) launch_one<0u>(k, grid, s);
    else if (ef == MCQ_CONV_SILU_OUT) launch_one<MCQ_CONV_SILU_OUT>(k, grid, s);
    else if (ef == MCQ_CONV_RESIDUAL) launch_one<MCQ_CONV_RESIDUAL>(k, grid, s);
    else if (ef == (MCQ_CONV_RESIDUAL | MCQ_CONV_DUAL_SILU)) launch_one<(MCQ_CONV_RESIDUAL | MCQ_CONV_DUAL_SILU)>(k, grid, s);
    else if (ef == MCQ_CONV_DUAL_SILU) launch_one<MCQ_CONV_DUAL_SILU>(k, grid, s);
    else if (ef == MCQ_CONV_SHUFFLE2) launch_one<MCQ_CONV_SHUFFLE2>(k, grid, s);
    else if ((ef & ~(unsigned)(MCQ_CONV_SILU_OUT | MCQ_CONV_RESIDUAL | MCQ_CONV_DUAL_SILU)) == 0u) launch_one<W16_RUNTIME>(k, grid, s);
    else return MCQ_EINVAL;                                    // (GDN / gate / multiplier epilogues stay with the direct form)
    return mcq_check_launch();
}
