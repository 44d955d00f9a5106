/*
 * mcquic_hip.h — C-ABI of libmcquic_hip.so, the MI355X (gfx950) replacement for the PyTorch ops
 * underneath McQuic's Compressor encode/decode hot path.
 *
 * The reference has no FFI for this path: it calls torch ops (nn.Conv2d / F.conv2d / torch.bmm /
 * argmin / advanced indexing).  Each entry point below cites the reference call site(s) it replaces
 * (paths relative to the reference tree).  The binding a reference maintainer would add (ctypes) is
 * shown in INTEGRATION.md and is what mcquic_amd/_lib.py does.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller; nothing is allocated inside;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); work is only enqueued,
 *     no entry point synchronises;
 *   - return 0 on success, a negative MCQ_E* code on invalid arguments; nothing throws;
 *   - tensors are dense fp32 NCHW; code indices are int64 [N, m, h, w]  (reference:
 *     mcquic/modules/quantizer.py:144-150, argmin returns int64).
 *   - all entry points are re-entrant.
 */
#ifndef MCQUIC_HIP_H
#define MCQUIC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a signature or struct in this header changes incompatibly (history in INTEGRATION.md section 2).  A binding
 * compares it with mcq_abi_version() of the library it loaded before calling anything else: a stale .so under new
 * prototypes (or the reverse) misaligns arguments silently otherwise.  3 = round 3 (mcq_rans_*_with_indexes take cdf_lens,
 * mcq_gate_f32 takes out_silu -- both changed in round 2 without a bump --, GroupNorm / logits-gradient entry points). */
#define MCQ_ABI_VERSION   9

#define MCQ_OK            0
#define MCQ_EINVAL       -1   /* NULL pointer / non-positive dimension / unsupported combination */
#define MCQ_ELAUNCH      -2   /* hipLaunchKernel failed (hipGetLastError() != hipSuccess)        */
#define MCQ_ETOOLARGE    -3   /* a per-image plane exceeds the 2 GiB buffer-addressing window    */

/* ---- conv flags (mcq_conv_desc.flags) ------------------------------------------------------- */
#define MCQ_CONV_SILU_IN    0x001u /* apply SiLU to the input on load   (mcquic/nn/blocks.py:70-78,179-200: act before conv) */
#define MCQ_CONV_SQUARE_IN  0x002u /* square the input on load          (mcquic/nn/gdn.py:75: F.conv2d(x ** 2, gamma, beta))  */
#define MCQ_CONV_SILU_OUT   0x004u /* apply SiLU to the result          (the act2 of _residulBlock, fused into conv1)        */
#define MCQ_CONV_RESIDUAL   0x008u /* y += res_scale * res              (blocks.py:77 `out += identity`; quantizer.py:318 `z - dequant`) */
#define MCQ_CONV_GDN        0x010u /* y = mul * (1/sqrt(acc))           (gdn.py:79  x * torch.rsqrt(std)) */
#define MCQ_CONV_IGDN       0x020u /* y = mul * sqrt(acc)               (gdn.py:91  x * torch.sqrt(std))  */
#define MCQ_CONV_GATE       0x040u /* y = mul * sigmoid(acc) + gate_id  (blocks.py:281-288 AttentionBlock.forward) */
#define MCQ_CONV_SHUFFLE2   0x080u /* store through nn.PixelShuffle(2)  (mcquic/nn/convs.py:221-255 pixelShuffle3x3) */
#define MCQ_CONV_MUL        0x200u /* y = mul * acc                     (GDN backward: 2 x * (gamma^T ds))                    */
#define MCQ_CONV_DSILU_MUL  0x400u /* y = acc * silu'(mul)              (backward of SiLU fused into the input-gradient conv:  */
                                   /*                                    mul = the SiLU's input; then + res as usual)           */
#define MCQ_CONV_GDN_BWD    0x4000u /* GDN backward in the epilogue of the recomputed s = beta + gamma x^2 launch (with SQUARE_IN): y = res / sqrt(s)
                                     * (the direct term dy f(s)), y_silu = res * mul * (-1/2 s^-3/2) (d s); res = dy, mul = x, RESIDUAL not set.
                                     * What torch.autograd derives for mcquic/nn/gdn.py:75-79; round 3 ran it as a launch of its own behind s. */
#define MCQ_CONV_IGDN_BWD   0x8000u /* the same for InvGenDivNorm (gdn.py:87-91): y = res * sqrt(s), y_silu = res * mul * (1/2 s^-1/2) */
#define MCQ_CONV_GATE_BWD   0x10000u /* (round 5) AttentionBlock gate backward in the epilogue of the recomputed s = conv1x1(b) launch (ksize 1, no other flag):
                                      * y = res * sigmoid(s) (d a), y_silu = res * mul * sigmoid(s) (1 - sigmoid(s)) (d s); res = d out, mul = a.
                                      * What torch.autograd derives for mcquic/nn/blocks.py:286-287; the forward keeps no s (the gate is the 1x1
                                      * launch's MCQ_CONV_GATE epilogue in training as in inference). */
#define MCQ_CONV_TAPS_LR    0x20000u /* (round 5) a PROMISE about w_packed, not an operation: the 3x3 stride-1 filter is zero outside its lower-right
                                      * 2 x 2 taps (dy, dx in {0, +1}) -- what mcq_pack_conv_dgrad_weight_f32 writes for a stride-2 layer
                                      * ([4 Cin, Cout, 3, 3] through MCQ_CONV_SHUFFLE2: torch.autograd's conv_transpose2d of mcquic/nn/convs.py:132-153
                                      * conv3x3(stride=2)).  The launch then walks 4 of the 9 taps; same sums.  No input prologue, no Winograd form. */
#define MCQ_CONV_POST_GDN   0x40000u  /* (round 6) the 1x1 layer that FOLLOWS this 3x3 convolution runs inside its launch, on the wave's fresh 128-channel
                                      * tile: v = conv(x) + bias, s = post_w @ v^2 + post_bias, y = v / sqrt(s)  (ResidualBlockWithStride: conv3x3 s2 -> GenDivNorm,
                                      * mcquic/nn/blocks.py:98-122, mcquic/nn/gdn.py:67-79).  Cout == 128, ksize 3; post_w from mcq_pack_post1x1_weight_f32. */
#define MCQ_CONV_POST_IGDN  0x80000u  /* the same with y = v * sqrt(s) (ResidualBlockShuffle: pixelShuffle3x3 -> InvGenDivNorm, blocks.py:141-159, gdn.py:81-91).
                                      * With MCQ_CONV_SHUFFLE2: Cout == 512 and w_packed / bias packed from the weight with its rows in SUB-PIXEL-MAJOR order
                                      * (row 128 T + c = conv channel 4 c + T: the 128 rows of tile T are the 128 shuffled channels of sub-pixel T = 2 dy + dx),
                                      * so that a wave holds all channels the normalisation mixes; the store goes to pixel (2 y + dy, 2 x + dx). */
#define MCQ_CONV_POST_GATE  0x100000u /* v = conv(x) + bias + res_scale * res, s = post_w @ v + post_bias, y = mul * sigmoid(s) + gate_id  (AttentionBlock:
                                      * the side stack's last convolution -> conv1x1 -> gate, blocks.py:281-288).  Cout == 128; DUAL_SILU allowed. */
#define MCQ_CONV_POST_MASK  (MCQ_CONV_POST_GDN | MCQ_CONV_POST_IGDN | MCQ_CONV_POST_GATE)
#define MCQ_CONV_DUAL_SILU  0x100u /* also store silu(y) to y_silu: the next block's act1(x), computed once per element */
#define MCQ_CONV_WINOGRAD2D 0x1000u /* OPT-IN like MCQ_CONV_WINOGRAD: F(2x2, 3x3), 4/9 of the multiplications; w_packed from   */
                                    /* mcq_pack_conv_weight_winograd2d_f32; Cout % 128 == 0, Cin % 8 == 0                                */
#define MCQ_CONV_WINOGRAD2D16 0x2000u /* OPT-IN: the same F(2x2, 3x3) arithmetic on v_mfma_f32_16x16x4_f32, two waves per SIMD (round 3);  */
                                     /* w_packed from mcq_pack_conv_weight_winograd16_f32; Cout % 128 == 0, Cin % 16 == 0; epilogues:    */
                                     /* SILU_OUT / RESIDUAL / DUAL_SILU combinations, or SHUFFLE2 alone                                   */
#define MCQ_CONV_WINOGRAD   0x800u /* OPT-IN, not the reference's arithmetic: 3x3 stride-1 layer in the Winograd F(2, 3) form along x */
                                   /* (2/3 of the multiplications, float32 throughout, results differ from the direct form in   */
                                   /* the last bits); w_packed then comes from mcq_pack_conv_weight_winograd_f32                 */

typedef struct mcq_conv_desc {
    const float* x;        /* [N, Cin, H, W]                                                      */
    const float* w_packed; /* from mcq_pack_conv_weight_f32                                        */
    const float* bias;     /* [Cout] or NULL                                                       */
    float*       y;        /* [N, Cout, Ho, Wo]; with SHUFFLE2: [N, Cout/4, 2Ho, 2Wo]              */
    float*       y_silu;   /* DUAL_SILU: same shape as y, receives silu(y)                         */
    const float* res;      /* RESIDUAL: same shape as y                                            */
    const float* mul;      /* GDN/IGDN/GATE: same shape as y                                       */
    const float* gate_id;  /* GATE: same shape as y                                                */
    int32_t N, Cin, H, W, Cout;
    int32_t ksize;         /* 1 or 3 (padding = ksize/2, zeros)                                    */
    int32_t stride;        /* 1 or 2                                                               */
    uint32_t flags;
    float   res_scale;     /* +1 or -1                                                             */
    int32_t tile;          /* 0 = auto; else (log2 split-K << 8) | (MB << 4) | NB forces the wave tile (testing / tuning);
                            * bit 0x400: the 128 x 64 tile of a 3x3 stride-1 layer over pixel PAIRS (same bits, wide epilogue accesses;
                            * even output width, flags within SiLU / twin / residual / silu' / PixelShuffle), bit 0x800: never */
    const float* post_w;    /* MCQ_CONV_POST_*: the following 1x1 layer's [128, 128] weight from mcq_pack_post1x1_weight_f32 (else NULL; ABI 9) */
    const float* post_bias; /* MCQ_CONV_POST_*: its bias [128] (GDN / IGDN: the folded beta) or NULL                                         */
} mcq_conv_desc;

/* (round 6) Operand stream of a [128, 128] 1x1 layer for the MCQ_CONV_POST_* epilogues: the contraction runs over the wave's OWN
 * accumulator registers, so its k-order is the accumulator order -- k-step t = 16 mb + r pairs channel 32 mb + (r & 3) + 8 (r >> 2)
 * (lanes 0..31) with that channel + 4 (lanes 32..63): out[(t * 64 + lane) * 4 + q] = w[32 q + (lane & 31)][that channel], followed by
 * a zero tail.  mcq_packed_post1x1_floats() floats. */
size_t mcq_packed_post1x1_floats(void);
int mcq_pack_post1x1_weight_f32(const float* w /* [128, 128] (1x1 OIHW) */, float* out, void* stream);
/* 0 if this geometry / flag set has no fused form (MCQ_CONV_POST_* in `flags`), else the number of 128 x 32 wave tiles the launch
 * would have: from 2048 (two per SIMD) mcq_conv2d_f32 takes it on its own; below that it refuses unless the caller forces the
 * unsplit tile (tile 0x41) -- small maps, whose 3x3 layer is otherwise split over waves, run the 1x1 layer as its own launch. */
int32_t mcq_conv2d_post_ok(int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t Cout, int32_t ksize, int32_t stride, uint32_t flags);

/* Number of floats mcq_pack_conv_weight_f32 writes for a [Cout, Cin, ks, ks] weight. */
size_t mcq_packed_conv_weight_floats(int32_t Cout, int32_t Cin, int32_t ksize);

/* Re-lay a dense OIHW weight (nn.Conv2d.weight, mcquic/nn/convs.py:77-100,257-276) into the MFMA operand streams the
 * conv kernels read: [Cout/128][Cin/2 x taps][64 lanes][4] followed by dense copies for the 64- and 32-row wave tiles
 * ([Cout/64][..][64][2], [Cout/32][..][64][1]), for 3x3 layers with Cin 64 / 128 and Cout % 16 == 0 (>= 32) a fourth copy in
 * the order of the small-launch kernel ([Cout/16][Cin/4 x 9 / 4][64][4], csrc/conv_t16.h) and, for 3x3 layers with <= 16 output
 * channels, the operand order of the 16-row image-head kernel; every copy zero padded (one launch). */
int mcq_pack_conv_weight_f32(const float* w_oihw, int32_t Cout, int32_t Cin, int32_t ksize,
                             float* w_packed, void* stream);

/* OPT-IN fast path (never the default): the operand stream of a 3x3 stride-1 layer in the Winograd F(2, 3) form along x --
 * per filter row (g0, g1, g2) the four transformed taps (g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2), formed in float64
 * and rounded once, in the same [Cout/128][Cin/2 x 12][64][4] (+ 64-row copy) order.  Used with MCQ_CONV_WINOGRAD, which
 * needs Cout % 64 == 0, ksize 3, stride 1 and no input prologue (SILU_IN / SQUARE_IN).  mcq_conv2d_winograd_ok tells
 * whether a geometry is taken (1) or would be refused (0). */
size_t mcq_packed_conv_winograd_floats(int32_t Cout, int32_t Cin);
/* The same in both directions, F(2x2, 3x3) (MCQ_CONV_WINOGRAD2D; Cout % 128 == 0 and Cin % 8 == 0): G g G^T, sixteen transformed taps per filter,
 * [Cout/32][Cin/2 x 16][64 lanes]; 4/9 of the multiplications. */
size_t mcq_packed_conv_winograd2d_floats(int32_t Cout, int32_t Cin);
int mcq_pack_conv_weight_winograd2d_f32(const float* w_oihw, int32_t Cout, int32_t Cin, float* w_packed, void* stream);
int mcq_pack_conv_weight_winograd_f32(const float* w_oihw, int32_t Cout, int32_t Cin, float* w_packed, void* stream);
/* ... and of the layer's stride-1 input-gradient convolution (torch.autograd's conv2d backward w.r.t. the input), straight from
 * the layer's own OIHW weight: w_packed holds mcq_packed_conv_winograd_floats(Cin, Cout) floats */
int mcq_pack_conv_dgrad_weight_winograd_f32(const float* w_oihw, int32_t Cout, int32_t Cin, float* w_packed, void* stream);
int32_t mcq_conv2d_winograd_ok(int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t Cout, int32_t ksize, int32_t stride, uint32_t flags);
/* 1 if mcq_conv2d_f32 / mcq_conv2d_multi_f32 (nprob problems) run this geometry and flag set on the small-launch kernel (16 x 16
 * tiles on v_mfma_f32_16x16x4_f32, one per workgroup: launches that would leave most of the GPU idle with 32-row tiles -- the
 * 4x4 ... 16x16 maps of a training step, the 12x8 / 24x16 levels of a single image), 0 if on the general one.  Same arithmetic,
 * same epilogue order; the summation order over (channel, tap) differs as between any two tile shapes. */
int32_t mcq_conv2d_small_launch(int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t Cout, int32_t ksize, int32_t stride, uint32_t flags,
                                int32_t nprob);

/* The operand stream of a layer's INPUT-GRADIENT convolution, packed straight from the layer's own OIHW weight
 * [Cout, Cin, k, k] in one launch (what torch.autograd derives for nn.Conv2d, mcquic/nn/convs.py:77-100):
 *   stride 1:          dX = conv(dY, W'),  W'[ci][co][tap] = W[co][ci][flipped tap]          -> a [Cin, Cout, k, k] conv
 *   stride 2 (3x3):    dX = PixelShuffle2(conv3x3(dY, W')), W' holding per input phase the taps that reach it
 *                                                                                            -> a [4 Cin, Cout, 3, 3] conv
 * mcq_dgrad_weight_shape gives that conv's (Cout_d, Cin_d); `out` holds mcq_packed_conv_weight_floats(Cout_d, Cin_d, k)
 * floats and is used with mcq_conv2d_f32 like any packed weight (MCQ_CONV_SHUFFLE2 for stride 2). */
int mcq_dgrad_weight_shape(int32_t Cout, int32_t Cin, int32_t ksize, int32_t stride, int32_t* Cout_d, int32_t* Cin_d);
int mcq_pack_conv_dgrad_weight_f32(const float* w, int32_t Cout, int32_t Cin, int32_t ksize, int32_t stride, float scale, float* out,
                                   void* stream);      /* every packed value times `scale` (GDN backward: 2 gamma^T; 1 otherwise) */

/* Dense 2-D convolution (zeros padding ksize/2) + fused prologue/epilogue.
 * Replaces nn.Conv2d.forward for conv3x3 / conv1x1 / pixelShuffle3x3 (mcquic/nn/convs.py:77-100,
 * 221-276), F.conv2d in GenDivNorm.forward (mcquic/nn/gdn.py:67-79), the SiLU / residual add of
 * _residulBlock.forward (mcquic/nn/blocks.py:70-78) and the gate of AttentionBlock.forward
 * (blocks.py:281-288). */
int mcq_conv2d_f32(const mcq_conv_desc* d, void* stream);

/* Up to mcq_conv2d_max_multi() independent convolutions of ONE geometry and flag set in one launch (descs[0..n): same N, Cin,
 * H, W, Cout, ksize, stride, flags, res_scale, tile; their own tensors and packed weights).  The two stacks of an
 * AttentionBlock (mcquic/nn/blocks.py:245-288) apply the same layer shapes to different tensors; on small maps, where a
 * launch is latency rather than work, they go out together. */
int32_t mcq_conv2d_max_multi(void);
int mcq_conv2d_multi_f32(const mcq_conv_desc* descs, int32_t n, void* stream);

/* GDN re-parametrisation folded once at load: out = max(p, bound)^2 - pedestal
 * (mcquic/nn/base.py:81-84 NonNegativeParametrizer.forward). */
int mcq_nonneg_reparam_f32(const float* p, float bound, float pedestal, float* out, int64_t n,
                           void* stream);
/* The same for up to mcq_nonneg_reparam_max_multi() parameters in ONE launch (host arrays of `count` entries): every GDN layer's
 * beta and gamma after an optimizer step (round 5; 20 launches per step of the qp = 2 model before). */
int32_t mcq_nonneg_reparam_max_multi(void);
int mcq_nonneg_reparam_multi_f32(const float* const* p, float* const* out, const int64_t* n, const float* bound, const float* pedestal,
                                 int32_t count, void* stream);

/* ---- multi-codebook quantizer ------------------------------------------------------------- */

/* Floats needed for the packed codebook of one level: the MFMA operand stream
 * m x [k/128][d/2][64][4] (+ prefetch tail) followed by the codeword norms c2 in operand layout. */
size_t mcq_packed_codebook_floats(int32_t m, int32_t k, int32_t d);

/* Pack codebook [m, k, d] into the operand stream and append c2[g, k] = sum_j c^2
 * (mcquic/modules/quantizer.py:163  c2 = (codebook ** 2).sum(-1)); padded codewords get +inf. */
int mcq_vq_pack_codebook_f32(const float* codebook, int32_t m, int32_t k, int32_t d,
                             float* cb_packed, void* stream);

/* codes[n, g, y, x] = argmin_k ( (|x_v|^2 + |c_k|^2) - 2 <x_v, c_k> ), first index on ties.
 * x: [N, m*d, h, w] (channel = g*d + j).  Replaces _multiCodebookQuantization._distance + encode
 * (mcquic/modules/quantizer.py:144-179) without materialising [N, m, h, w, k]. */
int mcq_vq_assign_f32(const float* x, const float* cb_packed, int64_t* codes,
                      int32_t N, int32_t m, int32_t d, int32_t h, int32_t w, int32_t k,
                      void* stream);
/* The same with a scratch buffer of mcq_vq_assign_workspace_bytes(...) bytes (0 = none needed; workspace may then be NULL):
 * launches with too few latent vectors to fill the GPU -- a single 768x512 image is 48 workgroups at the first level -- range
 * the codewords of a vector over up to 16 workgroups and fold the per-range minima in range order (first index on ties, as
 * within one workgroup).  Identical codes. */
size_t mcq_vq_assign_workspace_bytes(int32_t N, int32_t m, int32_t d, int32_t h, int32_t w, int32_t k);
int mcq_vq_assign_ws_f32(const float* x, const float* cb_packed, int64_t* codes,
                         int32_t N, int32_t m, int32_t d, int32_t h, int32_t w, int32_t k,
                         void* workspace, void* stream);

/* out[n, g*d + j, y, x] = codebook[g, codes[n, g, y, x], j]
 * (mcquic/modules/quantizer.py:249-259 _multiCodebookDeQuantization.decode).
 * Returns MCQ_OK; indices outside [0, k) are clamped (the reference would raise IndexError). */
int mcq_vq_gather_f32(const int64_t* codes, const float* codebook, float* out, float* out_silu /* or NULL */,
                      int32_t N, int32_t m, int32_t d, int32_t h, int32_t w, int32_t k,
                      void* stream);

/* ---- training-mode quantizer forward (BASELINE config #5, forward half) ------------------------------ */

/* logits[n, g, y, x, k] = ((-1 * dist) / sqrt(k)) * max(temperature[g], bound), dist as in mcq_vq_assign_f32.
 * Replaces _logit (mcquic/modules/quantizer.py:181-183) and the temperature scaling of _sample (:204). */
int mcq_vq_logits_f32(const float* x, const float* cb_packed, const float* temperature /* [m] */, float bound,
                      float* logits, int32_t N, int32_t m, int32_t d, int32_t h, int32_t w, int32_t k, void* stream);

/* Per latent vector, with the two uniform draws of the reference given as inputs u_drop, u_gumbel [N, m, h, w, k]:
 *   logits[c] += -1e9 where u_drop[c] ** drop_exponent[0] < freq_ema[g, c]      (_randomDrop, quantizer.py:194-200)
 *   codes        = argmax_c logits[c]                                            (forward, :232-239)
 *   sample_index = argmax_c softmax(logits + gumbel(u_gumbel))[c], sample_hot = (1 - s) + s with s the soft
 *   probability there: the value of y_hard - y_soft.detach() + y_soft, which is zero everywhere else
 *                                                                                (gumbelSoftmax, mcquic/nn/base.py:118-133) */
/* `rng_state` (round 4; NULL = both draws are given): {seed, offset} as two uint64 in DEVICE memory.  Where u_drop / u_gumbel is
 * NULL its draws are made inside the kernel from the state (stream 0 = the drop's, 1 = the Gumbel noise's): a counter-based
 * generator over (seed, offset, stream, element index), 24-bit uniforms in [0, 1) like torch.rand's -- the reference's
 * `torch.rand_like(logit)` tensors (2 x 134 MB at the first level of a training step) are never written or read.  The state is
 * read on the device, so a captured hipGraph sees the offset its host code advanced on the device before each replay. */
int mcq_vq_gumbel_sample_f32(float* logits, const float* u_drop /* or NULL */, const float* u_gumbel /* or NULL */,
                             const uint64_t* rng_state /* or NULL */, const float* freq_ema /* [m, k] */,
                             const float* drop_exponent /* device scalar */, int64_t* codes, int64_t* sample_index,
                             float* sample_hot, int64_t* code_counts /* [m, k] or NULL */, int32_t N, int32_t m, int32_t h, int32_t w,
                             int32_t k, void* stream);
/* `code_counts` (round 5): the level's code histogram -- what EntropyCoder.forward obtains by summing one-hot codes over images
 * and pixels (mcquic/modules/entropyCoder.py:33-35) -- is added to where each code is made (integer atomics: exact in any order);
 * the caller zeroes the buffer once per step (mcq_vq_step_prologue_f32). */
/* out[i] = the generator's draw for element i of `stream_id` under `rng_state` -- exactly what the two kernels around it use in
 * place of a NULL u_drop (stream 0) / u_gumbel (stream 1): for tests and for callers that want the tensors after all. */
int mcq_hash_uniform_f32(const uint64_t* rng_state, uint32_t stream_id, float* out, int64_t n, void* stream);

/* out[n, g*d + j, y, x] = sample_hot * codebook[g, sample_index, j]: bmm(sample, codebook) for the one-hot-valued
 * straight-through sample (_multiCodebookDeQuantization.forward, quantizer.py:262-274). */
int mcq_vq_dequant_soft_f32(const int64_t* sample_index, const float* sample_hot, const float* codebook, float* out,
                            float* out_silu /* or NULL: also silu(out), the consumer's first activation */,
                            int32_t N, int32_t m, int32_t d, int32_t h, int32_t w, int32_t k, void* stream);

/* out[n, g, y, x, k] = <x_v, c_k>: the logits kernel's GEMM without the distance / temperature transform
 * (backward: dSample = dDeq . C^T, the adjoint of quantizer.py:262-274). */
int mcq_vq_inner_f32(const float* x, const float* cb_packed, float* out, int32_t N, int32_t m, int32_t d, int32_t h, int32_t w,
                     int32_t k, void* stream);

/* Backward of gumbelSoftmax(hard=True) + _logit per latent vector: ds_inout holds dSample on entry and d dist on
 * exit; rowsum[v] = sum_k d dist, dtrow[v] = the vector's contribution to d max(temperature, bound).
 * `dlogits` (NULL = none): a gradient on the logits the forward returned (quantizer.py:232-239 returns them with their
 * graph); it is added in front of `_logit`.  The random drop's `+= -1e9` (quantizer.py:194-200) passes gradients through,
 * so dropped entries take part and `raw_logits` = the logits WITHOUT the drop (mcq_vq_logits_f32 again) must come along. */
int mcq_vq_softmax_bwd_f32(const float* logits, const float* u_gumbel /* or NULL: remade from rng_state */, const uint64_t* rng_state /* or NULL */,
                           float* ds_inout, const float* temperature, float bound,
                           float* rowsum, float* dtrow, const float* dlogits, const float* raw_logits,
                           int32_t N, int32_t m, int32_t h, int32_t w, int32_t k, void* stream);

/* Latent and codebook gradients of dist = |x|^2 + |c|^2 - 2 <x, c> given d dist, plus the codebook gradient of
 * sample @ codebook (rows sample_index scaled by sample_hot).  Deterministic (no atomics). */
int mcq_vq_soft_bwd_f32(const float* ddist, const float* rowsum, const float* x, const float* x_nhwc, const float* ddeq_nhwc,
                        const int64_t* sample_index, const float* sample_hot, const float* codebook, float* dx,
                        float* dcodebook, int32_t N, int32_t m, int32_t d, int32_t h, int32_t w, int32_t k, void* stream);

/* ---- per-step bookkeeping of the training quantizer (round 5; csrc/step_ops.hip) ------------------------------ */
#define MCQ_VQ_MAX_LEVELS 32   /* (8 until ABI 9; the reference's generator configs run 17 levels) */
/* the cap a binding chunks by (returns MCQ_VQ_MAX_LEVELS of the library it loaded) */
int32_t mcq_vq_max_levels(void);
/* ONE launch in front of the level cascade of a training forward (host arrays of `levels` <= MCQ_VQ_MAX_LEVELS entries):
 *   exponents[l] = -(log2 k_l - 1) * usage_l^2 + log2 k_l,  usage_l = mean(freq_ema_l > eps) clamped to [0, 1]
 *                  -- the exponent of _randomDrop (mcquic/modules/quantizer.py:194-198), rounded op by op like torch;
 *   rng_snaps[l] = {seed, offset + l} and the generator state {seed, offset} advanced by `levels` (both NULL: no generator);
 *   counts[0 .. counts_n) = 0  (the buffer mcq_vq_gumbel_sample_f32 adds the levels' code histograms into; NULL / 0: none).
 * Replaces seven tensor ops per level and the generator's clone + add. */
int mcq_vq_step_prologue_f32(const float* const* freq_ema /* [levels] device pointers to [m_l, k_l] */, const int32_t* m, const int32_t* k,
                             int32_t levels, float eps, float* exponents /* device [levels] */, uint64_t* rng_state /* device, or NULL */,
                             uint64_t* rng_snaps /* device [levels][2], or NULL */, int64_t* counts /* device, or NULL */, int64_t counts_n,
                             void* stream);
/* dtemperature[g] = mask_g * sum_{n, y, x} dtrow[n, g, y, x] with mask_g = (temperature[g] >= bound) | (sum < 0): the soft-max
 * backward's per-row temperature terms reduced in a fixed order, then LowerBound's gradient rule (mcquic/nn/base.py:24-29). */
int mcq_vq_temperature_grad_f32(const float* dtrow /* [N, m, hw] */, const float* temperature /* [m] */, float bound,
                                float* dtemperature /* [m] */, int32_t N, int32_t m, int32_t hw, void* stream);
/* freq_ema_l = (1 - ema) * counts_l / sum_k counts_l + ema * freq_ema_l for every level in ONE launch, in place
 * (EntropyCoder.forward, mcquic/modules/entropyCoder.py:28-44, after its all-reduce); `counts` = the levels' [m_l, k_l]
 * histograms back to back (mcq_vq_gumbel_sample_f32's code_counts). */
int mcq_freq_ema_update_f32(float* const* freq_ema /* [levels] device pointers */, const int32_t* m, const int32_t* k, int32_t levels,
                            const int64_t* counts, double ema /* a Python float on the reference side */, void* stream);
/* mcq_nonneg_reparam_bwd_f32 (below) for two parameters -- a GDN layer's beta [C] and gamma [C, C] -- in one launch. */
int mcq_nonneg_reparam_bwd2_f32(const float* p0, const float* dfolded0, float bound0, float* dp0, int64_t n0, const float* p1,
                                const float* dfolded1, float bound1, float* dp1, int64_t n1, void* stream);

/* ---- backward pass of the training step (BASELINE config #5) ----------------------------------------- */
/* Input gradients of the convolutions reuse mcq_conv2d_f32 with transformed weights (see mcquic_amd/autograd.py);
 * the reference obtains all of these from torch.autograd over nn.Conv2d / SiLU / GDN / sigmoid
 * (mcquic/nn/convs.py, nn/gdn.py:67-91, nn/blocks.py:70-78,281-288). */

/* Backward of mcq_nonneg_reparam_f32 under LowerBound's gradient rule (mcquic/nn/base.py:17-29,81-84):
 * g = 2 max(p, bound) dfolded;  dp = g where p >= bound or g < 0, else 0. */
int mcq_nonneg_reparam_bwd_f32(const float* p, const float* dfolded, float bound, float* dp, int64_t n, void* stream);

/* out[n][p][c] = x[n][c][p] (squared if `square`): channel-major copies feeding the weight-gradient GEMM. */
int mcq_nchw_to_nhwc_f32(const float* x, float* out, int32_t N, int32_t C, int32_t HW, int32_t square, void* stream);
/* The same for the two operands of one weight-gradient GEMM (x [N,Cx,HWx], squared if `square_x`, and y [N,Cy,HWy]) in
 * a single launch. */
int mcq_nchw_to_nhwc_pair_f32(const float* x, float* x_out, int32_t Cx, int32_t HWx, int32_t square_x, const float* y,
                              float* y_out, int32_t Cy, int32_t HWy, int32_t N, void* stream);

/* dW[co][ci][ky][kx] = sum_{n,yo,xo} dY[n][co][yo][xo] * X[n][ci][yo*stride + ky - k/2][xo*stride + kx - k/2]
 * from NHWC copies of X [N,H,W,Cin] and dY [N,Ho,Wo,Cout]; `workspace` holds the per-range partial sums.  If `dbias` is
 * not NULL it also receives db[co] = sum_{n,yo,xo} dY[n][co][yo][xo] (the bias gradient of the same conv, summed by the
 * waves that stream dY anyway). */
size_t mcq_conv2d_wgrad_workspace_floats(int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t Cout, int32_t ksize,
                                         int32_t stride);
int mcq_conv2d_wgrad_f32(const float* x_nhwc, const float* dy_nhwc, float* dw, float* dbias, float* workspace, int32_t N,
                         int32_t Cin, int32_t H, int32_t W, int32_t Cout, int32_t ksize, int32_t stride, void* stream);

/* The same gradient for 3x3 stride-1 convolutions straight from the NCHW tensors X [N,Cin,H,W] and dY [N,Cout,H,W], W a
 * multiple of 8, each tensor below 1 GiB (mcq_conv2d_wgrad_nchw_workspace_floats returns 0 for any other shape: use the
 * NHWC entry point then).  No channel-major copies: a wave walks an 8-pixel strip down the rows with a three-row window of
 * X in registers, all nine taps per dY operand (csrc/wgrad_rows.hip).  Deterministic.
 * Shapes with N H W <= 512 pixels, H W a multiple of 4 (<= 256) and channel counts in sixteens -- the 4x4 / 8x8 maps of a training
 * step -- take a one-pass kernel on 16 x 16 tiles instead (csrc/wgrad_t16.h: whole images staged in LDS, no partial sums): the
 * workspace query returns 1 for them (as for the other small maps) and `workspace` is not touched.  The same holds for the 1x1
 * entry point below. */
size_t mcq_conv2d_wgrad_nchw_workspace_floats(int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t Cout);
int mcq_conv2d_wgrad_nchw_f32(const float* x, const float* dy, float* dw, float* dbias, float* workspace, int32_t N, int32_t Cin,
                              int32_t H, int32_t W, int32_t Cout, void* stream);
/* `nconv` (1 .. mcq_conv2d_wgrad_nchw_max_group()) convolutions of ONE shape in one launch pair: x[c], dy[c] -> dw[c], dbias[c]
 * (dbias NULL, or NULL entries, for none).  The pointer tables are host arrays read during the call; `workspace` holds
 * nconv * mcq_conv2d_wgrad_nchw_workspace_floats(...) floats.  The 16x16 ... 8x8 levels of a training step are
 * launch-bound: a block's weight gradients go out together (mcquic_amd/autograd.py). */
int32_t mcq_conv2d_wgrad_nchw_max_group(void);
int mcq_conv2d_wgrad_nchw_group_f32(const float* const* x, const float* const* dy, float* const* dw, float* const* dbias,
                                    int32_t nconv, float* workspace, int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t Cout,
                                    void* stream);

/* The 3x3 stride-2 case (ResidualBlockWithStride's convs, the 3-channel stem): X [N,Cin,H,W] with H, W even, dY
 * [N,Cout,H/2,W/2], W/2 a multiple of 8.  The walk runs over dY's rows with a ring of four input rows. */
size_t mcq_conv2d_wgrad_s2_nchw_workspace_floats(int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t Cout);
int mcq_conv2d_wgrad_s2_nchw_f32(const float* x, const float* dy, float* dw, float* dbias, float* workspace, int32_t N, int32_t Cin,
                                 int32_t H, int32_t W, int32_t Cout, void* stream);

/* The 1x1 case (the AttentionBlock gate conv; with square_x the gamma of GDN / IGDN, whose operand is x^2,
 * mcquic/nn/gdn.py:75): dW[co][ci] = sum dY[co] * X[ci] straight from NCHW, H even, W a multiple of 8.  The workspace query
 * returns 0 for shapes this kernel does not take. */
size_t mcq_conv2d_wgrad1x1_nchw_workspace_floats(int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t Cout);
int mcq_conv2d_wgrad1x1_nchw_f32(const float* x, const float* dy, float* dw, float* dbias, float* workspace, int32_t N, int32_t Cin,
                                 int32_t H, int32_t W, int32_t Cout, int32_t square_x, void* stream);

/* out[c] = sum_{n,p} x[n][c][p]   (bias / beta gradients); optional workspace of min(N, 16) * C floats. */
int mcq_channel_sum_f32(const float* x, float* out, float* workspace, int32_t N, int32_t C, int32_t HW, void* stream);

/* Stand-alone forms of ops that the inference path fuses into conv prologues / epilogues; the training graph keeps
 * them separate so that each has its own backward:  y = silu(x);  out = a * sigmoid(b) + x;  out = alpha a + beta b. */
int mcq_silu_f32(const float* x, float* y, int64_t n, void* stream);
int mcq_gate_f32(const float* a, const float* b, const float* x, float* out, float* out_silu /* or NULL */, int64_t n, void* stream);
int mcq_axpby_f32(const float* a, const float* b, float alpha, float beta, float* out, float* out_silu /* or NULL */, int64_t n,
                  void* stream);

/* dx = dy * d/dx silu(x) (+ other: the gradient reaching x along a second path, e.g. a strided block's skip convolution,
 * mcquic/nn/blocks.py:98-122 -- the sum torch.autograd's engine would make with a launch of its own). */
int mcq_silu_bwd_f32(const float* x, const float* dy, const float* other /* or NULL */, float* dx, int64_t n, void* stream);

/* AttentionBlock gate out = a * sigmoid(b) + x: da = dout * s, db = dout * a * s (1 - s). */
int mcq_gate_bwd_f32(const float* a, const float* b, const float* dout, float* da, float* db, int64_t n, void* stream);

/* GDN / IGDN y = x * f(s): dx_direct = dy * f(s), ds = dy * x * f'(s) (f = s^-1/2, or s^1/2 when `inverse`). */
int mcq_gdn_bwd_prep_f32(const float* x, const float* s, const float* dy, int32_t inverse, float* dx_direct, float* ds,
                         int64_t n, void* stream);

/* out[n][c*4 + i*2 + j][y][x] = in[n][c][2y + i][2x + j]: the adjoint of nn.PixelShuffle(2). */
int mcq_pixel_unshuffle2_f32(const float* in, float* out, int32_t N, int32_t C, int32_t H, int32_t W, void* stream);

/* ---- small element-wise helpers on the path ------------------------------------------------ */

/* out = a + b  (quantizer.py:354  xHat = q + sideHead(formerLevel)). */
int mcq_add_f32(const float* a, const float* b, float* out, float* out_silu /* or NULL */, int64_t n, void* stream);
/* out = (a + b) + c: the three gradient paths that meet at the input of an AttentionBlock (main stack, side stack, the gate's
 * identity term; what torch.autograd accumulates with two additions for mcquic/nn/blocks.py:281-288) in one launch. */
int mcq_add3_f32(const float* a, const float* b, const float* c, float* out, int64_t n, void* stream);

/* mean((a - b)^2) over n floats -> out[0] (F.mse_loss, the distortion term of mcquic/loss/__init__.py:62), and its gradient
 * da = 2 (a - b) / n * dloss[0] (db = -da when non-NULL).  Two launches, a fixed summation order (double partials in
 * `workspace`, mcq_mse_workspace_bytes(n) bytes) and no memset: safe inside a captured hipGraph, which a library reduction that
 * zeroes semaphores with hipMemsetAsync is not on ROCm 7.2 ("graph replays"). */
size_t mcq_mse_workspace_bytes(int64_t n);
int mcq_mse_f32(const float* a, const float* b, float* out, void* workspace, int64_t n, void* stream);
int mcq_mse_bwd_f32(const float* a, const float* b, const float* dloss, float* da, float* db /* or NULL */, int64_t n, void* stream);
/* Gradient clipping by global norm over ONE flat buffer (mcquic/train/trainer.py:280 `self._optimizer.clip_grad_norm(4.0)`;
 * torch.nn.utils.clip_grad_norm_'s arithmetic): out[0] = sum(x^2) with the reduction above (workspace: mcq_mse_workspace_bytes(n)),
 * then x *= max_norm / (sqrt(sumsq) + eps) where that factor is < 1; the norm goes to norm_out[0] when non-NULL.  Everything
 * stays on the device: no host read between the two launches, so both are captured with the optimizer's update. */
int mcq_sumsq_f32(const float* x, float* out, void* workspace, int64_t n, void* stream);
int mcq_clip_by_norm_f32(float* x, const float* sumsq, float max_norm, float eps, float* norm_out /* or NULL */, int64_t n, void* stream);

/* Many tensors -> their slices of one flat buffer in ONE launch (the gradients of a captured training step into the buffer the
 * all-reduce and the optimizer read; torch.cat moves 666 tensors in six launches): device tables as for mcq_adam_step_f32 --
 * `src_ptrs` [ntensors] addresses, `dst_offsets` / `numel` [ntensors] in floats, and one (tensor, first element) entry per
 * mcq_adam_chunk()-element chunk in `blk_tensor` / `blk_first` [nblocks]. */
int mcq_gather_flat_f32(const void* src_ptrs, float* flat, const int64_t* dst_offsets, const int64_t* numel, const int32_t* blk_tensor,
                        const int64_t* blk_first, int32_t nblocks, void* stream);

/* Adam / AdamW over a whole model in ONE launch (the `self._optimizer.step()` of mcquic/train/trainer.py:283 with the reference's
 * `Adam`, configs/a800_8.yaml:20-25; arithmetic of torch.optim.Adam / AdamW: lerp of exp_avg, mul + addcmul of exp_avg_sq, bias
 * corrections from the step count, param -= lr / bc1 * exp_avg / (sqrt(exp_avg_sq) / sqrt(bc2) + eps); L2 or decoupled decay).
 * All lists are DEVICE arrays: ptr_tables = uint64[4][ntensors] (param, grad, exp_avg, exp_avg_sq addresses), numel[ntensors],
 * and one (blk_tensor, blk_first) entry per mcq_adam_chunk()-element chunk of every tensor (nblocks entries).  `step` is a device
 * float the call increments; `lr_dev` (device float) overrides `lr` when non-NULL, so a scheduled rate needs no re-capture;
 * `scalars` = 16 bytes of device scratch.  Two launches, nothing read by the host. */
int32_t mcq_adam_chunk(void);
int mcq_adam_step_f32(const void* ptr_tables, int32_t ntensors, const int64_t* numel, const int32_t* blk_tensor, const int64_t* blk_first,
                      int32_t nblocks, float* step, const float* lr_dev /* or NULL */, double lr, double beta1, double beta2, double eps,
                      double weight_decay, int32_t decoupled, int32_t maximize, void* scalars, void* stream);

/* u8 = trunc(clamp(((x + 1) / 2) * 255.999, 0, 255))   (mcquic/utils/vision.py:143-146 DeTransform). */
int mcq_detransform_u8(const float* x, uint8_t* out, int64_t n, void* stream);

/* ---- entropy coder beside the tensor path (host code, no GPU work) ------------------------------ */

/* Up to mcq_pack_conv_weight_max_multi() weights of ONE shape packed in one launch: w[i] -> out[i], forward streams
 * (`dgrad` = 0) or input-gradient streams (`dgrad` = 1, with `stride` and `scale` as in mcq_pack_conv_dgrad_weight_f32).  Layers
 * with <= 16 output channels (they carry a second copy) are refused: pack those one by one.  After an optimizer step every
 * convolution of the network re-packs both streams; this turns 660 launches into ~45. */
int32_t mcq_pack_conv_weight_max_multi(void);
int mcq_pack_conv_weight_multi_f32(const float* const* w, float* const* out, int32_t n, int32_t Cout, int32_t Cin, int32_t ksize,
                                   int32_t dgrad, int32_t stride, float scale, void* stream);
/* The same with a section mask per weight (NULL or 0 = all): bit 0 the 128-row copy of the 3x3 operand stream, bit 1 the 64-row
 * copy, bit 2 the 32-row copy, bit 3 the 16x16-tile order; copies not named are left as they are.  For a re-pack INSIDE a captured
 * training step, whose launches -- and therefore the copies they read -- never change: mcq_conv_section_trace(1) starts recording
 * which copy every conv launch reads per packed buffer (clearing earlier records), (0) stops; mcq_conv_sections_used(packed) =
 * the recorded mask of a buffer (0: not seen).  1x1 weights ignore the mask.  (A training loop's re-pack after optimizer.step():
 * train/trainer.py's step; nothing in the reference corresponds to the mask -- cuDNN keeps no re-laid copies.) */
int mcq_pack_conv_weight_multi_masked_f32(const float* const* w, float* const* out, const uint8_t* masks, int32_t n, int32_t Cout,
                                          int32_t Cin, int32_t ksize, int32_t dgrad, int32_t stride, float scale, void* stream);
void mcq_conv_section_trace(int32_t on);
uint32_t mcq_conv_sections_used(const float* packed);

/* cdf[0..k] (uint32, cdf[0] = 0, cdf[k] = 1 << precision, strictly increasing) from pmf[0..k-1].
 * Same arithmetic as pmfToQuantizedCDF (third_party/CompressAI/cpp_exts/ops.cpp:42-111). */
int mcq_pmf_to_quantized_cdf(const float* pmf, int32_t k, int32_t precision, uint32_t* cdf);

/* One rANS stream over symbols[0..n): symbol i is coded with CDF number indexes[i]; CDF c occupies
 * cdfs[cdf_starts[c] ...] with cdf_lens[c] entries actually present; cdf_sizes[c] follows the reference's convention
 * (sentinel slot = cdf_sizes[c] - 2; the reference's caller passes k + 2 over k + 1 entries,
 * mcquic/modules/entropyCoder.py:121), offsets[c] is subtracted from the symbol.  A symbol whose table slot would lie
 * beyond cdf_lens[c] is MCQ_EINVAL (the reference reads past the table there).  Returns the number of bytes written to
 * `out`, or a negative MCQ_E* (MCQ_ETOOLARGE if `capacity` is too small: 4 * n + 8 always suffices without bypass
 * symbols).  Bit-identical to RansEncoder.encodeWithIndexes (cpp_exts/buffered_rans_encoder.cpp:104-196 over
 * ryg_rans/rans64.h). */
int64_t mcq_rans_encode_with_indexes(const int32_t* symbols, const int32_t* indexes, int64_t n,
                                     const uint32_t* cdfs, const int32_t* cdf_starts, const int32_t* cdf_sizes,
                                     const int32_t* cdf_lens, const int32_t* offsets, int32_t n_cdfs, uint8_t* out, int64_t capacity);

/* Inverse of the above (cpp_exts/rans_decoder.cpp:104-167); MCQ_EINVAL on a truncated / malformed stream. */
int mcq_rans_decode_with_indexes(const uint8_t* in, int64_t nbytes, const int32_t* indexes, int64_t n,
                                 const uint32_t* cdfs, const int32_t* cdf_starts, const int32_t* cdf_sizes,
                                 const int32_t* cdf_lens, const int32_t* offsets, int32_t n_cdfs, int32_t* out_symbols);

/* All images of one level in ONE call (replaces the reference's Python loop `for code in codes: for codePerImage in
 * code: encoder.encodeWithIndexes(...)`, mcquic/modules/entropyCoder.py:108-126): n_streams independent streams of n
 * symbols each (symbols[n_streams][n], one shared `indexes[n]`), stream i written to out + i * stride (stride >= 4 n + 8
 * without bypass symbols) with its length in out_nbytes[i].  Streams are spread over `n_threads` host threads (<= 0: one
 * per hardware thread); the bytes of a stream do not depend on the thread count.  Returns MCQ_OK or the first failing
 * stream's error. */
int mcq_rans_encode_batch_with_indexes(const int32_t* symbols, int64_t n_streams, int64_t n, const int32_t* indexes,
                                       const uint32_t* cdfs, const int32_t* cdf_starts, const int32_t* cdf_sizes,
                                       const int32_t* cdf_lens, const int32_t* offsets, int32_t n_cdfs, uint8_t* out,
                                       int64_t stride, int64_t* out_nbytes, int32_t n_threads);

/* Inverse (entropyCoder.py:141-154): stream i = in[in_offsets[i] .. in_offsets[i + 1]), decoded into out_symbols[i][n]. */
int mcq_rans_decode_batch_with_indexes(const uint8_t* in, const int64_t* in_offsets, int64_t n_streams, const int32_t* indexes,
                                       int64_t n, const uint32_t* cdfs, const int32_t* cdf_starts, const int32_t* cdf_sizes,
                                       const int32_t* cdf_lens, const int32_t* offsets, int32_t n_cdfs, int32_t* out_symbols,
                                       int32_t n_threads);

/* ---- validation metrics (callers of the path: mcquic/validate/handlers.py:14-41) --------------------------------
 * MS-SSIM of two uint8 batches x, y [N, C, H, W] as the reference's MsSSIM handler computes it
 * (handlers.py:14-27 -> metrics.py:69-104 `_ssim`, :142-193 `ms_ssim`, module defaults metrics.py:222: 11-tap sigma-1.5
 * window, K = (0.01, 0.03), data range 255, five levels with weights metrics.py:19).  out[N] receives the MS-SSIM VALUE
 * in [0, 1] (the reference module returns 1 - value and the handler prints -10 log10 of that).  H and W must exceed 160
 * (metrics.py:163-166), else MCQ_EINVAL.  `workspace` holds the pooled pyramids and partial sums:
 * mcq_ms_ssim_workspace_bytes(N, C, H, W) bytes (0 for an invalid shape), 8-byte aligned, contents undefined after. */
size_t mcq_ms_ssim_workspace_bytes(int32_t N, int32_t C, int32_t H, int32_t W);
int mcq_ms_ssim_u8(const uint8_t* x, const uint8_t* y, float* out, void* workspace, int32_t N, int32_t C, int32_t H,
                   int32_t W, void* stream);
/* Host helper: the 11 float32 window taps the kernels use (metrics.py:22-37 for size 11, sigma 1.5). */
void mcq_ms_ssim_window(float* out11);

/* out[n] = sum over the per_image bytes of image n of (x - y)^2, exact (int64).  The reference's PSNR
 * (metrics.py:264-274) is 10 log10(255^2 / (out[n] / per_image + 1e-4)) in float64. */
int mcq_sqdiff_sum_u8(const uint8_t* x, const uint8_t* y, int64_t* out, int64_t per_image, int32_t N, void* stream);

/* F(2x2, 3x3) weights for MCQ_CONV_WINOGRAD2D16: U = G g G^T (float64, rounded once) in the operand order of the 16 x 16 x 4
 * instance, [Cout/32][Cin/4][16 positions][2 halves][64 lanes] + one group of zeros. */
size_t mcq_packed_conv_winograd16_floats(int32_t Cout, int32_t Cin);
int mcq_pack_conv_weight_winograd16_f32(const float* w, int32_t Cout, int32_t Cin, float* out, void* stream);

/* ---- the weight gradients' reduce passes, batched (round 5) ----------------------------------------------------------------------------
 * mcq_wgrad_defer(1): from now on the mcq_conv2d_wgrad*_nchw* entry points record their second pass (the fixed-order sum of the
 * partial tiles in `workspace` -> dW in OIHW order, db) instead of launching it; mcq_wgrad_flush(0, stream) launches everything
 * recorded, 80 convolutions per launch, and empties the record (discard != 0: empties it without launching, after an error).
 * Until the flush the caller keeps every workspace alive and reads no dW / db.  Same sums in the same order as the one-by-one pass.
 * mcq_wgrad_pending(): convolutions recorded and not yet flushed ON THE CALLING THREAD'S CURRENT DEVICE.  The switch is process-wide
 * (the autograd engine records from its own per-device thread): one backward pass at a time uses it; the record is kept per
 * device (ABI 9), a flush launches the current device's jobs on `stream` and a discard empties every device's.
 * The caller also keeps every dW / db alive until the flush (a dropped one would be written over whatever took its memory). */
void mcq_wgrad_defer(int32_t on);
int32_t mcq_wgrad_pending(void);
int mcq_wgrad_flush(int32_t discard, void* stream);

/* ---- GroupNorm (`denseNorm=True`) ------------------------------------------------------------------------------------
 * y = (x - mean) * rstd * gamma[c] + beta[c] over each (image, group) of C / groups adjacent channels, biased variance,
 * rstd = 1 / sqrt(var + eps): nn.GroupNorm(groups, C) as the reference's ResidualBlock inserts it in place of its second
 * activation when denseNorm is set (mcquic/nn/blocks.py:179-200).  gamma / beta may be NULL (1 / 0).  y_silu (NULL = none)
 * receives silu(y).  mean_out / rstd_out [N * groups] (both or neither): what the backward pass needs. */
/* `workspace` (round 5; mcq_group_norm_workspace_floats floats, 0 = none needed, NULL = take the one-workgroup-per-run kernel):
 * planes of 256 pixels and more are cut into 8192-float chunks, one workgroup each, whose (mean, M2) pairs are merged in
 * (plane, chunk) order -- Neon's GroupNorm(32, 32) on 512 x 512 maps is 32 chunks per run instead of one workgroup. */
size_t mcq_group_norm_workspace_floats(int32_t N, int32_t C, int32_t HW, int32_t groups);
int mcq_group_norm_f32(const float* x, const float* gamma, const float* beta, float* y, float* y_silu, float* mean_out,
                       float* rstd_out, float* workspace /* or NULL */, int32_t N, int32_t C, int32_t HW, int32_t groups, float eps,
                       void* stream);
/* Backward of the above: dx [N, C, HW], dgamma / dbeta [C] (NULL = not wanted) from x, dy and the forward's mean / rstd.
 * workspace: mcq_group_norm_bwd_workspace_floats(N, C, HW, groups) floats.  Deterministic (no atomics). */
size_t mcq_group_norm_bwd_workspace_floats(int32_t N, int32_t C, int32_t HW, int32_t groups);
int mcq_group_norm_bwd_f32(const float* x, const float* dy, const float* gamma, const float* mean, const float* rstd, float* dx,
                           float* dgamma, float* dbeta, float* workspace, int32_t N, int32_t C, int32_t HW, int32_t groups,
                           void* stream);

/* Library / build identification: returns a static string "mcquic_hip <ver> gfx950". */
/* Launches a kernel with an invalid configuration on purpose and returns what every entry point returns when its launch is
 * refused: MCQ_ELAUNCH.  Test hook for the error path (tests/test_gpu_ops.py); harmless (launch errors are not sticky). */
int mcq_selftest_launch_failure(void* stream);

const char* mcq_version(void);
/* MCQ_ABI_VERSION the library was built from. */
int32_t mcq_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MCQUIC_HIP_H */
