// Multi-codebook vector quantizer kernels for gfx950.
//
// mcq_vq_assign_f32 replaces _multiCodebookQuantization._distance + encode
// (reference: mcquic/modules/quantizer.py:144-179): per group g and latent vector v
//     dist[v, k] = (|x_v|^2 + |c_k|^2) - 2 <x_v, c_k>,   code[v] = argmin_k dist[v, k] (first index on ties)
// The reference materialises inter = bmm(x, codebook^T) as [N, m, h, w, k]; here the inner products
// stay in MFMA accumulators (rows = 128 codewords, cols = 64 latent vectors per wave), the epilogue
// of every 128-codeword tile folds them into a per-lane running (min, argmin), and only the int64
// codes are written.  The epilogue keeps the reference's rounding sequence (x2 + c2) - 2*inter.
//
// |c_k|^2 is needed in the accumulator's row layout.  It is obtained bit-exactly with one extra
// MFMA per 32 rows: A[i][0] = c2[i], B[0][j] = 1, everything else 0  =>  D[i][j] = c2[i].
#include "mcq_common.h"
#include "vq_common.h"
#include "../../include/mcquic_hip.h"
#include <math.h>

#ifndef VQ_TWO_LEVEL
#define VQ_TWO_LEVEL 0          // build switch: 1 = the tile epilogue first folds a band's 16 distances to their minimum (v_min3_f32) and runs the
                                // compare / select pairs only when some lane's band beats its running minimum.  Built and measured in round 6
                                // (profiles/r06_vq_two_level.txt): same codes, 1-5 % SLOWER on every shape (config #4 3.05 vs 3.02 ms, qp=2 level 0
                                // 0.922 vs 0.906 ms, d = 16 0.955 vs 0.922 ms) -- the wave-uniform skip costs the epilogue its overlap with the
                                // next tile's MFMAs and 38 % of the bands take both paths.  Left off.
#endif

namespace {

// SPC = compile-time k-steps per tile (p.Sp) for the common vector lengths, 0 = run-time loop.  With the k-steps of a
// tile fully unrolled the whole tile body is straight-line code and hipcc counts the prefetch ring exactly
// (`s_waitcnt vmcnt(N)` per step); around a run-time inner loop it merges the loop-entry and back-edge states
// conservatively and drains the ring at every loop head (112 -> 117 TFLOP/s on config #4).
// XLDS (d = 256): a wave re-reads its 64 vectors (64 KB) for each of its codeword tiles; that does not fit L1, and
// taking it from L2 every time cost 7 % (ablation: activation loads forced to one hot line 122 -> 130 TFLOP/s).  In this
// mode the four waves of a workgroup are the four codeword slices of ONE vector tile, which they stage in LDS once
// ([channel][64 vectors]: the B operand of a k-step is then two conflict-free ds_read_b32 per lane).
template <int SPC, bool XLDS>
__global__ __launch_bounds__(256, 2) void vq_assign_kernel(VqK p) {
    __shared__ float xs[XLDS ? 2 * SPC * 64 : 1];
    constexpr int MB = VQ_MB, NB = VQ_NB, PF = VQ_PF;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // 4 waves per workgroup = (4 >> cs_log2) vector tiles x (1 << cs_log2) slices of the codeword tiles
    const int CS = 1 << p.cs_log2;
    const int slice = wave & (CS - 1);
    const int gw = blockIdx.x * (4 >> p.cs_log2) + (wave >> p.cs_log2);
    const bool active = gw * NB < p.total_blocks;            // wave-uniform
    if (CS == 1 && !active) return;                          // (sliced waves stay for the barrier)
    const int g = blockIdx.y;
    const int hi = lane >> 5, j = lane & 31;
    const int BW = 1 << p.bw_log2;
    const int ly = j >> p.bw_log2, lx = j & (BW - 1);
    const int BH = 32 >> p.bw_log2;
    const int HW = p.h * p.w;
    const unsigned group_bytes = (unsigned)p.d * (unsigned)HW * 4u;

    int img[NB], yo[NB], xo[NB];
    bool valid[NB];
    unsigned pixoff[NB];
    __amdgpu_buffer_rsrc_t rsrc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        int pb = gw * NB + nb;
        const bool pbv = pb < p.total_blocks;
        if (!pbv) pb = p.total_blocks - 1;
        const int per_img = p.nby * p.nbx;
        const int n = pb / per_img;
        const int rem = pb - n * per_img;
        const int by = rem / p.nbx;
        const int bx = rem - by * p.nbx;
        img[nb] = n;
        yo[nb] = by * BH + ly;
        xo[nb] = bx * BW + lx;
        valid[nb] = pbv && yo[nb] < p.h && xo[nb] < p.w;
        pixoff[nb] = valid[nb] ? (unsigned)(yo[nb] * p.w + xo[nb]) * 4u : MCQ_OOB;
        // the descriptor covers exactly this image's group-g channels, so padded k-steps read 0
        rsrc[nb] = mcq_make_rsrc(mcq_uniform_ptr(p.x + ((size_t)n * p.m + g) * (size_t)p.d * HW), group_bytes);
    }

    if (XLDS) {
        // wave w stages channels w, w + 4, ... of the tile (its own lanes' pixels: every slice has the same geometry)
        for (int c = wave; c < 2 * SPC; c += 4) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                if (hi == 0) xs[c * 64 + nb * 32 + j] = mcq_buffer_load(rsrc[nb], pixoff[nb] + (unsigned)c * (unsigned)HW * 4u);
        }
        __syncthreads();
    }

    // |x_v|^2, sequential over the d channels of the group (both half-waves compute it redundantly)
    // (sixteen independent loads per batch, both blocks together; the additions keep the channel order)
    float x2[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) x2[nb] = 0.0f;
    for (int c0 = 0; c0 < p.d; c0 += 16) {
        float v[NB][16];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {    // channels past d are out of range = 0: they add +0
                // (XLDS, round 6: from the staged tile -- four waves reading their 64 KB again from global memory made the launch pull
                //  3.2x its algorithmic bytes through the fabric, profiles/r05_pmc_by_kernel_vq.txt; same values, same order)
                if (XLDS) v[nb][i] = c0 + i < 2 * SPC ? xs[(c0 + i) * 64 + nb * 32 + j] : 0.0f;
                else v[nb][i] = mcq_buffer_load(rsrc[nb], pixoff[nb] + (unsigned)(c0 + i) * (unsigned)HW * 4u);
            }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) x2[nb] = x2[nb] + v[nb][i] * v[nb][i];
    }

    f32x4v A[PF];
    float B[PF][NB];
    // (grid.z workgroups x CS waves share the codeword tiles of a vector tile: small launches, see the launcher)
    const int per_slice = (p.ntile + CS * p.zs - 1) / (CS * p.zs);
    const int tile0r = ((int)blockIdx.z * CS + slice) * per_slice;
    const int tile0 = tile0r < p.ntile ? tile0r : p.ntile;             // this wave's codeword tiles: [tile0, tile1) (possibly none)
    const int tile1 = active ? (tile0 + per_slice < p.ntile ? tile0 + per_slice : p.ntile) : tile0;
    // operand addresses stay off the vector ALU inside the k-loop (a VALU instruction between two MFMAs costs the
    // matrix pipe ~10 cycles, tools/probes/mfma_issue.hip): per-lane offsets are loop constants, the running part is
    // a wave-uniform soffset advanced by the scalar unit
    const __amdgpu_buffer_rsrc_t wr = mcq_make_rsrc(p.cbp + ((size_t)g * p.ntile + tile0) * p.Sp * 256, 0x7fffffffu);
    const unsigned wlane = (unsigned)lane * 16u;
    unsigned wso = 0;
    const f32x4v* c2l = reinterpret_cast<const f32x4v*>(p.c2p) + ((size_t)g * (p.ntile + 1) + tile0) * 64 + lane;
    int ls = 0;
    unsigned soffL = 0;
    const unsigned step_bytes = 2u * (unsigned)HW * 4u;
    unsigned voffL[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) voffL[nb] = valid[nb] ? pixoff[nb] + (unsigned)(hi * HW) * 4u : MCQ_OOB;

    auto issue = [&](int st) {
        A[st] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(wr, (int)wlane, (int)wso, 0));
        wso += 1024;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            B[st][nb] = XLDS ? xs[(2 * ls + hi) * 64 + nb * 32 + j] : mcq_buffer_load_s(rsrc[nb], voffL[nb], soffL);
        ++ls;
        soffL += step_bytes;
        if (ls == p.Sp) { ls = 0; soffL = 0; }
    };

    float best[NB];
    int bidx[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { best[nb] = INFINITY; bidx[nb] = 0; }
    const float bone = hi == 0 ? 1.0f : 0.0f;

    f32x4v c2a = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
    if (tile0 < tile1) {
#pragma unroll
        for (int st = 0; st < PF; ++st) issue(st);
        c2a = c2l[0];
    }

    for (int tile = tile0; tile < tile1; ++tile) {
        const f32x4v c2n = c2l[(size_t)(tile - tile0 + 1) * 64];   // next tile's norms (one spare tile is allocated)
        f32x16 acc[MB][NB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;

        auto step = [&](int st, int t) {
            // fully unrolled variants: the slot being refilled belongs to k-step (t + PF) % SPC of the tile -- a
            // compile-time number, so the LDS read takes an immediate offset and nothing is computed per lane
            const int lsn = SPC ? (t + PF) % (SPC ? SPC : 1) : ls;
            const unsigned son = SPC ? (unsigned)lsn * step_bytes : soffL;
            // vector-block major, each activation slot refilled right after its last use (as in conv_mfma_kernel)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[st][mb], B[st][nb], acc[mb][nb], 0, 0, 0);
                B[st][nb] = XLDS ? xs[(2 * lsn + hi) * 64 + nb * 32 + j] : mcq_buffer_load_s(rsrc[nb], voffL[nb], son);
            }
            A[st] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(wr, (int)wlane, (int)wso, 0));
            wso += 1024;
            ++ls;
            soffL += step_bytes;
            if (ls == p.Sp) { ls = 0; soffL = 0; }
            // keep the software pipeline as written (without the fence hipcc regroups the loads of the body)
            __builtin_amdgcn_sched_barrier(0);
        };
        if (SPC) {
#pragma unroll
            for (int t = 0; t < SPC; t += PF) {
#pragma unroll
                for (int st = 0; st < PF; ++st) step(st, t + st);
            }
        } else {
            for (int t = 0; t < p.Sp; t += PF) {
#pragma unroll
                for (int st = 0; st < PF; ++st) step(st, 0);
            }
        }

        const int word0 = tile * 128;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.0f;
            const f32x16 c2d = __builtin_amdgcn_mfma_f32_32x32x2f32(c2a[mb], bone, z, 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                // (x2 + c2) - 2 * inter ; 2 * inter is exact, so the fused form rounds identically.  Two distances per
                // instruction (v_pk_add_f32 / v_pk_fma_f32: same IEEE operations, half the issue slots)
                float dv[16];
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2v s2 = f32x2v{x2[nb], x2[nb]} + f32x2v{c2d[r], c2d[r + 1]};
                    const f32x2v dv2 = __builtin_elementwise_fma(f32x2v{-2.0f, -2.0f}, f32x2v{acc[mb][nb][r], acc[mb][nb][r + 1]}, s2);
                    dv[r] = dv2[0]; dv[r + 1] = dv2[1];
                }
                if (VQ_TWO_LEVEL) {
                    // (round 6, off: see VQ_TWO_LEVEL) two levels: the band's 16 distances are first folded to their minimum (v_min3_f32: eight instructions),
                    // and the compare / select pair per distance that finds WHICH row it was only runs when some lane of the wave has a
                    // band that beats its running minimum -- past the first tiles that is rare (a band improves a lane's minimum with
                    // probability ~ 1 / bands seen).  Same result: no distance below the running minimum <=> band minimum not below it,
                    // and inside the slow path the rows are visited in index order with a strict comparison, as before.
                    float t0 = fminf(fminf(dv[0], dv[1]), dv[2]), t1 = fminf(fminf(dv[3], dv[4]), dv[5]), t2 = fminf(fminf(dv[6], dv[7]), dv[8]),
                          t3 = fminf(fminf(dv[9], dv[10]), dv[11]), t4 = fminf(fminf(dv[12], dv[13]), dv[14]);
                    const float tmin = fminf(fminf(fminf(t0, t1), t2), fminf(fminf(t3, t4), dv[15]));
                    if (__builtin_amdgcn_ballot_w64(tmin < best[nb]) == 0ull) continue;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int word = word0 + mb * 32 + mcq_drow(r, hi);
                    if (dv[r] < best[nb]) { best[nb] = dv[r]; bidx[nb] = word; }
                }
            }
        }
        c2a = c2n;
    }

    // the two half-waves hold interleaved codeword rows of the same latent vector
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const float ob = __shfl_xor(best[nb], 32);
        const int oi = __shfl_xor(bidx[nb], 32);
        if (ob < best[nb] || (ob == best[nb] && oi < bidx[nb])) { best[nb] = ob; bidx[nb] = oi; }
    }
    if (CS > 1) {
        // slices meet in LDS; slice 0 folds the others in slice order -- a later slice holds larger codeword indices,
        // so it only wins with a strictly smaller distance (first index on ties, like torch.argmin)
        __shared__ float s_best[4][NB][64];
        __shared__ int s_idx[4][NB][64];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            s_best[wave][nb][lane] = best[nb];
            s_idx[wave][nb][lane] = bidx[nb];
        }
        __syncthreads();
        if (slice != 0 || !active) return;
        for (int s = 1; s < CS; ++s) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const float ob = s_best[wave + s][nb][lane];
                const int oi = s_idx[wave + s][nb][lane];
                if (ob < best[nb]) { best[nb] = ob; bidx[nb] = oi; }
            }
        }
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
        if (hi == 0 && valid[nb]) {
            const size_t v = (((size_t)img[nb] * p.m + g) * p.h + yo[nb]) * p.w + xo[nb];
            if (p.zs == 1) p.codes[v] = (int64_t)bidx[nb];
            else {
                const size_t nvec = (size_t)p.N * p.m * p.h * p.w;
                p.ws_best[(size_t)blockIdx.z * nvec + v] = best[nb];
                p.ws_idx[(size_t)blockIdx.z * nvec + v] = bidx[nb];
            }
        }
}

// the ranges of a vector meet: range order = codeword order, so a later range only wins with a strictly smaller distance
// (first index on ties, like torch.argmin and like the slices inside a workgroup)
__global__ void vq_fold_kernel(const float* __restrict__ ws_best, const int* __restrict__ ws_idx, int zs, size_t nvec,
                               int64_t* __restrict__ codes) {
    const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nvec) return;
    float best = ws_best[v];
    int idx = ws_idx[v];
    for (int z = 1; z < zs; ++z) {
        const float ob = ws_best[(size_t)z * nvec + v];
        if (ob < best) { best = ob; idx = ws_idx[(size_t)z * nvec + v]; }
    }
    codes[v] = (int64_t)idx;
}

// codebook [m, k, d] -> cbp [m][ntile][Sp][64][4] (+ zero tail) and c2p [m][ntile + 1][64][4]
__global__ void vq_pack_kernel(const float* __restrict__ cb, int m, int k, int d, int Sp, int ntile,
                               float* __restrict__ cbp, size_t cb_total, size_t cb_alloc) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cb_alloc) return;
    float v = 0.0f;
    if (i < cb_total) {
        const int q = (int)(i & 3);
        const int lane = (int)((i >> 2) & 63);
        size_t rest = i >> 8;
        const int s = (int)(rest % Sp); rest /= Sp;
        const int tile = (int)(rest % ntile);
        const int g = (int)(rest / ntile);
        const int word = tile * 128 + 32 * q + (lane & 31);
        const int c = 2 * s + (lane >> 5);
        if (word < k && c < d) v = cb[((size_t)g * k + word) * d + c];
    }
    cbp[i] = v;
}

__global__ void vq_c2_kernel(const float* __restrict__ cb, int m, int k, int d, int ntile, float* __restrict__ c2p,
                             size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int q = (int)(i & 3);
    const int lane = (int)((i >> 2) & 63);
    size_t rest = i >> 8;
    const int tile = (int)(rest % (ntile + 1));
    const int g = (int)(rest / (ntile + 1));
    const int word = tile * 128 + 32 * q + (lane & 31);
    float v = 0.0f;
    if ((lane >> 5) == 0) {
        if (tile < ntile && word < k) {
            const float* row = cb + ((size_t)g * k + word) * d;
            float s = 0.0f;
            for (int c = 0; c < d; ++c) s = s + row[c] * row[c];
            v = s;
        } else {
            v = INFINITY;   // padded codewords can never win the argmin
        }
    }
    c2p[i] = v;
}

__global__ void vq_gather_kernel(const int64_t* __restrict__ codes, const float* __restrict__ cb, float* __restrict__ out,
                                 float* __restrict__ out2, int N, int m, int d, int hw, int k) {
    // one thread per (n, g, pixel); writes d channel planes (coalesced across threads)
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)N * m * hw;
    if (i >= total) return;
    const int pix = (int)(i % hw);
    const size_t ng = i / hw;
    const int g = (int)(ng % m);
    const size_t n = ng / m;
    int64_t code = codes[i];
    code = code < 0 ? 0 : (code >= k ? k - 1 : code);
    const float* row = cb + ((size_t)g * k + (size_t)code) * d;
    float* o = out + ((n * m + g) * (size_t)d) * hw + pix;
    for (int c = 0; c < d; ++c) o[(size_t)c * hw] = row[c];
    if (out2) {
        float* o2 = out2 + ((n * m + g) * (size_t)d) * hw + pix;
        for (int c = 0; c < d; ++c) o2[(size_t)c * hw] = mcq_silu(row[c]);
    }
}

__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                           float* __restrict__ out2, int64_t n) {
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        const f32x4v va = *reinterpret_cast<const f32x4v*>(a + i);
        const f32x4v vb = *reinterpret_cast<const f32x4v*>(b + i);
        const f32x4v vs = va + vb;
        *reinterpret_cast<f32x4v*>(out + i) = vs;
        if (out2) *reinterpret_cast<f32x4v*>(out2 + i) = f32x4v{mcq_silu(vs[0]), mcq_silu(vs[1]), mcq_silu(vs[2]), mcq_silu(vs[3])};
    } else {
        for (; i < n; ++i) {
            const float v = a[i] + b[i];
            out[i] = v;
            if (out2) out2[i] = mcq_silu(v);
        }
    }
}

// out = (a + b) + c, the rounding order of two chained additions (the three gradient paths that meet at an AttentionBlock's input)
__global__ void add3_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c, float* __restrict__ out,
                            int64_t n) {
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        const f32x4v va = *reinterpret_cast<const f32x4v*>(a + i);
        const f32x4v vb = *reinterpret_cast<const f32x4v*>(b + i);
        const f32x4v vc = *reinterpret_cast<const f32x4v*>(c + i);
        *reinterpret_cast<f32x4v*>(out + i) = (va + vb) + vc;
    } else {
        for (; i < n; ++i) out[i] = (a[i] + b[i]) + c[i];
    }
}

__global__ void detransform_kernel(const float* __restrict__ x, uint8_t* __restrict__ out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        // (x - min) / (max - min) with min = -1, max = 1, then * (255 + 1.0 - 1e-3), clamp, truncate
        float v = (x[i] - (-1.0f)) / 2.0f;
        v = v * 255.999f;
        v = fminf(fmaxf(v, 0.0f), 255.0f);
        out[i] = (uint8_t)v;
    }
}

}  // namespace

extern "C" size_t mcq_packed_codebook_floats(int32_t m, int32_t k, int32_t d) {
    if (m <= 0 || k <= 0 || d <= 0) return 0;
    const size_t ntile = (size_t)(k + 127) / 128;
    const size_t cb = ((size_t)m * ntile * vq_sp(d) + VQ_PF) * 256;   // operand stream + prefetch tail
    const size_t c2 = (size_t)m * (ntile + 1) * 256;                  // norms (+1 spare tile per group)
    return cb + c2;
}

extern "C" int mcq_vq_pack_codebook_f32(const float* codebook, int32_t m, int32_t k, int32_t d, float* cb_packed,
                                        void* stream) {
    if (!codebook || !cb_packed || m <= 0 || k <= 0 || d <= 0) return MCQ_EINVAL;
    const int ntile = (k + 127) / 128, Sp = vq_sp(d);
    const size_t cb_total = (size_t)m * ntile * Sp * 256;
    const size_t cb_alloc = cb_total + (size_t)VQ_PF * 256;
    const size_t c2_total = (size_t)m * (ntile + 1) * 256;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(vq_pack_kernel, dim3((unsigned)((cb_alloc + 255) / 256)), dim3(256), 0, s, codebook, m, k, d, Sp,
                       ntile, cb_packed, cb_total, cb_alloc);
    hipLaunchKernelGGL(vq_c2_kernel, dim3((unsigned)((c2_total + 255) / 256)), dim3(256), 0, s, codebook, m, k, d, ntile,
                       cb_packed + cb_alloc, c2_total);
    return mcq_check_launch();
}

namespace {
// Workgroups per vector tile (grid.z) for launches that would leave most of the GPU idle -- one 768x512 image is 24 vector
// tiles x 2 codebooks at the first level: 48 workgroups walking 8192 codewords each, 139 us for 20 us of MFMA work.  The
// codeword tiles are then ranged over up to 16 workgroups (each at least one tile per wave) until ~1536 waves exist.
int vq_assign_ranges(long long vtiles, int m, int ntile, int cs_log2) {
    const long long waves = vtiles * m << cs_log2;
    int zs = 1;
    while (zs < 16 && waves * zs < 1536 && (ntile >> cs_log2) >= 2 * zs) zs *= 2;
    return zs;
}
int vq_assign_plan(int N, int m, int d, int h, int w, int k, int& cs_log2_out) {
    int bw_log2;
    block_shape(h, w, bw_log2);
    const int bw = 1 << bw_log2, bh = 32 >> bw_log2;
    const long long tb = (long long)N * ((w + bw - 1) / bw) * ((h + bh - 1) / bh);
    const long long vtiles = (tb + VQ_NB - 1) / VQ_NB;
    const int ntile = (k + 127) / 128, Sp = vq_sp(d);
    int cs_log2 = 0;
    double best_cost = 1e30;
    for (int lg = 0; lg <= 2 && (1 << lg) <= ntile; ++lg) {
        const long long waves = vtiles * m << lg;
        const double cost = (double)((waves + 2047) / 2048) / (double)(1 << lg);
        if (cost < best_cost - 1e-12) { best_cost = cost; cs_log2 = lg; }
    }
    if (Sp == 128 && ntile >= 4) cs_log2 = 2;          // the LDS-staged mode: four slices share one vector tile
    cs_log2_out = cs_log2;
    return vq_assign_ranges(vtiles, m, ntile, cs_log2);
}
}  // namespace

extern "C" size_t mcq_vq_assign_workspace_bytes(int32_t N, int32_t m, int32_t d, int32_t h, int32_t w, int32_t k) {
    if (N <= 0 || m <= 0 || d <= 0 || h <= 0 || w <= 0 || k <= 0) return 0;
    int cs;
    const int zs = vq_assign_plan(N, m, d, h, w, k, cs);
    return zs > 1 ? (size_t)zs * N * m * h * w * 8u : 0;
}

extern "C" int mcq_vq_assign_f32(const float* x, const float* cb_packed, int64_t* codes, int32_t N, int32_t m, int32_t d,
                                 int32_t h, int32_t w, int32_t k, void* stream) {
    return mcq_vq_assign_ws_f32(x, cb_packed, codes, N, m, d, h, w, k, nullptr, stream);
}

extern "C" int mcq_vq_assign_ws_f32(const float* x, const float* cb_packed, int64_t* codes, int32_t N, int32_t m, int32_t d,
                                    int32_t h, int32_t w, int32_t k, void* workspace, void* stream) {
    if (!x || !cb_packed || !codes || N <= 0 || m <= 0 || d <= 0 || h <= 0 || w <= 0 || k <= 0) return MCQ_EINVAL;
    if ((uint64_t)d * h * w * 4ull >= 0x80000000ull) return MCQ_ETOOLARGE;
    VqK p;
    p.x = x; p.codes = codes; p.N = N; p.m = m; p.d = d; p.h = h; p.w = w; p.k = k;
    p.ntile = (k + 127) / 128;
    p.Sp = vq_sp(d);
    const size_t cb_alloc = ((size_t)m * p.ntile * p.Sp + VQ_PF) * 256;
    p.cbp = cb_packed;
    p.c2p = cb_packed + cb_alloc;
    block_shape(h, w, p.bw_log2);
    const int bw = 1 << p.bw_log2, bh = 32 >> p.bw_log2;
    p.nbx = (w + bw - 1) / bw;
    p.nby = (h + bh - 1) / bh;
    const long long tb = (long long)N * p.nbx * p.nby;
    if (tb > 0x7fffffffLL) return MCQ_ETOOLARGE;
    p.total_blocks = (int)tb;
    // Every wave walks all codeword tiles of its 64 vectors, so the launch has (vector tiles x m) waves whatever k is:
    // too few for the 2048 wave slots (2 per SIMD) on the small levels, and 1.5 rounds for config #4.  Splitting the
    // codeword tiles over 2 or 4 waves of a workgroup multiplies the waves and divides their length; pick the split
    // with the fewest whole rounds of work.
    const long long vtiles = (tb + VQ_NB - 1) / VQ_NB;
    int cs_log2 = 0;
    const int zs_plan = vq_assign_plan(N, m, d, h, w, k, cs_log2);
    p.cs_log2 = cs_log2;
    p.zs = workspace ? zs_plan : 1;                          // (without a workspace: one workgroup per vector tile, as before)
    const size_t nvec = (size_t)N * m * h * w;
    p.ws_best = static_cast<float*>(workspace);
    p.ws_idx = reinterpret_cast<int*>(p.ws_best + (size_t)p.zs * nvec);
    const int per_wg = 4 >> cs_log2;
    const unsigned gx = (unsigned)((vtiles + per_wg - 1) / per_wg);
    const dim3 grid(gx, (unsigned)m, (unsigned)p.zs);
    if (p.Sp == 32) hipLaunchKernelGGL((vq_assign_kernel<32, false>), grid, dim3(256), 0, (hipStream_t)stream, p);          // d = 64
    else if (p.Sp == 8) hipLaunchKernelGGL((vq_assign_kernel<8, false>), grid, dim3(256), 0, (hipStream_t)stream, p);      // d = 16 (model No. 12), d = 8 ... 15
    else if (p.Sp == 128 && cs_log2 == 2) hipLaunchKernelGGL((vq_assign_kernel<128, true>), grid, dim3(256), 0, (hipStream_t)stream, p);   // d = 256
    else if (p.Sp == 128) hipLaunchKernelGGL((vq_assign_kernel<128, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((vq_assign_kernel<0, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
    if (p.zs > 1)
        hipLaunchKernelGGL(vq_fold_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p.ws_best, p.ws_idx, p.zs,
                           nvec, codes);
    return mcq_check_launch();
}

extern "C" int mcq_vq_gather_f32(const int64_t* codes, const float* codebook, float* out, float* out_silu, int32_t N,
                                 int32_t m, int32_t d, int32_t h, int32_t w, int32_t k, void* stream) {
    if (!codes || !codebook || !out || N <= 0 || m <= 0 || d <= 0 || h <= 0 || w <= 0 || k <= 0) return MCQ_EINVAL;
    const size_t total = (size_t)N * m * h * w;
    hipLaunchKernelGGL(vq_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, codes,
                       codebook, out, out_silu, N, m, d, h * w, k);
    return mcq_check_launch();
}

extern "C" int mcq_add_f32(const float* a, const float* b, float* out, float* out_silu, int64_t n, void* stream) {
    if (!a || !b || !out || n <= 0) return MCQ_EINVAL;
    const int64_t threads = (n + 3) / 4;
    hipLaunchKernelGGL(add_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, out_silu, n);
    return mcq_check_launch();
}

extern "C" int mcq_add3_f32(const float* a, const float* b, const float* c, float* out, int64_t n, void* stream) {
    if (!a || !b || !c || !out || n <= 0) return MCQ_EINVAL;
    const int64_t threads = (n + 3) / 4;
    hipLaunchKernelGGL(add3_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, c, out, n);
    return mcq_check_launch();
}

extern "C" int mcq_detransform_u8(const float* x, uint8_t* out, int64_t n, void* stream) {
    if (!x || !out || n <= 0) return MCQ_EINVAL;
    hipLaunchKernelGGL(detransform_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, out, n);
    return mcq_check_launch();
}

extern "C" const char* mcq_version(void) { return "mcquic_hip 0.3.0 gfx950"; }
extern "C" int32_t mcq_abi_version(void) { return MCQ_ABI_VERSION; }
