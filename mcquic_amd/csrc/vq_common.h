// Shared host/device definitions of the vector-quantizer kernels (vq.hip, vq_train.hip).
#pragma once
#include <stdint.h>

namespace {

#ifndef MCQ_VQ_PF
#define MCQ_VQ_PF 8
#endif
constexpr int VQ_MB = 4, VQ_NB = 2, VQ_PF = MCQ_VQ_PF;     // wave tile 128 codewords x 64 vectors; prefetch ring depth in k-steps

struct VqK {
    const float* x; const float* cbp; const float* c2p; int64_t* codes;
    int N, m, d, h, w, k;
    int Sp;            // k-steps (channel pairs) per tile, padded to a multiple of VQ_PF
    int ntile;         // 128-codeword tiles
    int bw_log2, nbx, nby, total_blocks;
    int cs_log2;       // vq_assign: 1 << cs_log2 waves of a workgroup share one vector tile, each a slice of the codeword tiles
    int zs;            // vq_assign: workgroups (grid.z) that share one vector tile, each a range of the codeword tiles (1 = none)
    float* ws_best;    // zs > 1: [zs][N m h w] best distance / its codeword per range, folded by vq_fold_kernel
    int* ws_idx;
};

inline void block_shape(int Ho, int Wo, int& lg_out) {
    int best_log2 = 5; double best_util = -1.0;
    for (int lg = 5; lg >= 2; --lg) {
        const int bw = 1 << lg, bh = 32 >> lg;
        const double cover = (double)((Ho + bh - 1) / bh * bh) * (double)((Wo + bw - 1) / bw * bw);
        const double util = (double)Ho * Wo / cover;
        if (util > best_util + 1e-9) { best_util = util; best_log2 = lg; }
    }
    lg_out = best_log2;
}

inline int vq_sp(int d) { return (((d + 1) / 2) + VQ_PF - 1) / VQ_PF * VQ_PF; }


// fill the geometry / operand-stream fields common to every VQ launch; returns false when the shape is too large
inline bool vq_setup(VqK& p, const float* x, const float* cb_packed, int N, int m, int d, int h, int w, int k) {
    p.x = x; p.N = N; p.m = m; p.d = d; p.h = h; p.w = w; p.k = k;
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
    if (tb > 0x7fffffffLL) return false;
    p.total_blocks = (int)tb;
    return true;
}

}  // namespace
