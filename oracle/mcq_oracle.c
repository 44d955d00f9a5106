/*
 * mcq_oracle.c -- plain-C CPU restatement of the two contractions on McQuic's Compressor path.
 *
 * TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).  Nothing under
 * mcquic_amd/ links or calls this.  It is an implementation-independent second opinion next to the PyTorch
 * restatement in oracle/mcquic_ref.py (which is the one pinned bit-for-bit against the real reference): naive
 * loops, fp32 data, sequential fp32 or fp64 accumulation -- no BLAS, no oneDNN.
 *
 *   mcq_oracle_conv2d      nn.Conv2d(k, stride, padding=k/2, zeros) + bias
 *                          (reference: mcquic/nn/convs.py:77-100,257-276)
 *   mcq_oracle_vq_assign   dist = (x2 + c2) - 2*inter, argmin first index
 *                          (reference: mcquic/modules/quantizer.py:144-179); also reports the fp64 gap between the
 *                          best and the second-best codeword, which the parity tests use for the near-tie audit
 *   mcq_oracle_vq_gather   codebook[g, code] -> [n, m*d, h, w]   (quantizer.py:249-259)
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

/* x [N,Cin,H,W], w [Cout,Cin,ks,ks], bias [Cout] or NULL, y [N,Cout,Ho,Wo]; accumulate in double if wide != 0 */
int mcq_oracle_conv2d(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int H, int W,
                      int Cout, int ks, int stride, int wide) {
    if (!x || !w || !y || (ks != 1 && ks != 3) || (stride != 1 && stride != 2)) return -1;
    const int pad = ks / 2;
    const int Ho = (H + 2 * pad - ks) / stride + 1, Wo = (W + 2 * pad - ks) / stride + 1;
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Cout; ++co)
            for (int yo = 0; yo < Ho; ++yo)
                for (int xo = 0; xo < Wo; ++xo) {
                    double accd = 0.0;
                    float accf = 0.0f;
                    for (int ci = 0; ci < Cin; ++ci)
                        for (int ky = 0; ky < ks; ++ky)
                            for (int kx = 0; kx < ks; ++kx) {
                                const int yi = yo * stride + ky - pad, xi = xo * stride + kx - pad;
                                if (yi < 0 || yi >= H || xi < 0 || xi >= W) continue;
                                const float a = x[(((size_t)n * Cin + ci) * H + yi) * W + xi];
                                const float b = w[(((size_t)co * Cin + ci) * ks + ky) * ks + kx];
                                if (wide) accd += (double)a * (double)b; else accf = fmaf(a, b, accf);
                            }
                    float v = wide ? (float)accd : accf;
                    if (bias) v += bias[co];
                    y[(((size_t)n * Cout + co) * Ho + yo) * Wo + xo] = v;
                }
    return 0;
}

/* x [N, m*d, h, w], codebook [m, k, d] -> codes int64 [N, m, h, w]; gap (may be NULL) [N, m, h, w] double */
int mcq_oracle_vq_assign(const float* x, const float* cb, int64_t* codes, double* gap, int N, int m, int d, int h,
                         int w, int k) {
    if (!x || !cb || !codes) return -1;
    const size_t hw = (size_t)h * w;
    for (int n = 0; n < N; ++n)
        for (int g = 0; g < m; ++g)
            for (size_t p = 0; p < hw; ++p) {
                const float* xv = x + ((size_t)n * m + g) * d * hw + p;       /* stride hw between components */
                float x2 = 0.0f;
                for (int j = 0; j < d; ++j) x2 += xv[j * hw] * xv[j * hw];
                float best = INFINITY; int64_t bi = 0;
                double best64 = INFINITY, second64 = INFINITY;
                for (int c = 0; c < k; ++c) {
                    const float* cv = cb + ((size_t)g * k + c) * d;
                    float c2 = 0.0f, inter = 0.0f;
                    double d64 = 0.0;
                    for (int j = 0; j < d; ++j) {
                        c2 += cv[j] * cv[j];
                        inter = fmaf(xv[j * hw], cv[j], inter);
                        const double df = (double)xv[j * hw] - (double)cv[j];
                        d64 += df * df;
                    }
                    const float dist = (x2 + c2) - 2.0f * inter;                /* the reference's rounding order */
                    if (dist < best) { best = dist; bi = c; }
                    if (d64 < best64) { second64 = best64; best64 = d64; } else if (d64 < second64) second64 = d64;
                }
                const size_t o = ((size_t)n * m + g) * hw + p;
                codes[o] = bi;
                if (gap) gap[o] = second64 - best64;
            }
    return 0;
}

int mcq_oracle_vq_gather(const int64_t* codes, const float* cb, float* out, int N, int m, int d, int h, int w, int k) {
    if (!codes || !cb || !out) return -1;
    const size_t hw = (size_t)h * w;
    for (int n = 0; n < N; ++n)
        for (int g = 0; g < m; ++g)
            for (size_t p = 0; p < hw; ++p) {
                const int64_t c = codes[((size_t)n * m + g) * hw + p];
                if (c < 0 || c >= k) return -2;
                for (int j = 0; j < d; ++j) out[(((size_t)n * m + g) * d + j) * hw + p] = cb[((size_t)g * k + c) * d + j];
            }
    return 0;
}
