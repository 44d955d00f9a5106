// Weight gradient of the 3x3 stride-1 convolutions straight from the NCHW tensors (training step, BASELINE config #5).
//
//   dW[co][ci][dy][dx] = sum_{n, y, x} dY[n][co][y][x] * X[n][ci][y + dy - 1][x + dx - 1]        (zero padding)
//
// which the reference gets from torch.autograd over nn.Conv2d (mcquic/nn/convs.py:77-100).  A GEMM whose reduction axis
// is the pixel axis: D[co][ci] += A[co][pixel] * B[pixel][ci] on v_mfma_f32_32x32x2_f32, two pixels per instruction
// (lane half kh = lane >> 5 supplies pixel kh of the pair, lane & 31 the channel).
//
// Why no NHWC copies (csrc/train_ops.hip's first version made one pair per conv): a sum over pixels may visit them in any
// order, so a lane reads FOUR consecutive pixels of its own channel row with one 16-byte load -- lane half 0 the pixels
// x0 .. x0+3 of an 8-pixel strip, half 1 the pixels x0+4 .. x0+7 -- and element q of that vector is k-step q's operand
// on both sides.  A wave walks a strip down the rows of an image and keeps a window of three input rows (6 values per
// lane and row: the 16-byte load plus the two neighbours) in registers: the nine taps of a k-step are nine MFMAs on the
// same dY operand and nine different window registers -- 36 MFMAs per row for four loads (the forward kernel needs
// three loads per eight).  Horizontal padding is a per-lane out-of-range offset fixed for the whole walk, vertical padding a
// wave-uniform out-of-range soffset (scalar select): no vector ALU work in the loop except the bias sums.
//   wave        = (32 co x 32 ci tile, all 9 taps: 144 accumulator registers) x (a range of strip-rows)
//   workgroup   = 4 waves of one tile, summed through LDS in a fixed tree before the partials leave the CU
//   second pass = wgrad_rows_reduce_kernel: fixed-order sum over the workgroups' partials -> dW in OIHW order, db
// Deterministic: no atomics anywhere.
#include <algorithm>
#include <mutex>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "mcq_common.h"
#include "../../include/mcquic_hip.h"

#ifndef MCQ_WGRAD_MAX_GROUP
#define MCQ_WGRAD_MAX_GROUP 16
#endif
#ifndef MCQ_WGROWS_WAVES
#define MCQ_WGROWS_WAVES 2048          // resident waves aimed at (2 per SIMD)
#endif
#ifndef MCQ_WGROWS_MIN_ROWS
#define MCQ_WGROWS_MIN_ROWS 8          // strip-rows a wave should at least own (prologue + partial-sum cost)
#endif

namespace {

constexpr unsigned ROWS_OOB_S = 0x40000000u;     // soffset that puts every lane out of range (tensors are < 1 GiB here)

constexpr int ROWS_MAX_CONVS = MCQ_WGRAD_MAX_GROUP;    // convolutions of one shape per launch (pointer pairs travel as kernel arguments)

struct WgRowsK {
    const float* x[ROWS_MAX_CONVS]; const float* dy[ROWS_MAX_CONVS]; float* part; float* bias_part;
    int nconv, groups;               // part[conv][group][tap][co][ci], bias_part[conv][group * 4][co]
    int N, Cin, Cout, H, W;
    int strips, row_chunks, rpc;     // W / 8; row ranges per (image, strip); rows per range
    int units, splits, ups;          // (image, strip, row range) walks; waves per tile; walks per wave
};

struct RowsPlan { int F, strips, row_chunks, rpc, units, splits, ups, groups, gmax; };

inline bool rows_plan_search(int N, int Cin, int H, int W, int Cout, RowsPlan& r, int taps, bool stride2, int nconv);

// (the plan of a shape is looked up far more often than it changes: the last few are remembered per host thread)
inline bool rows_plan(int N, int Cin, int H, int W, int Cout, RowsPlan& r, int taps = 9, bool stride2 = false, int nconv = 1) {
    struct Memo { int key[8]; RowsPlan plan; bool ok, used; };
    static thread_local Memo memo[16];
    static thread_local int next = 0;
    const int key[8] = {N, Cin, H, W, Cout, taps, stride2 ? 1 : 0, nconv};
    for (const Memo& m : memo) {
        bool same = m.used;
        for (int i = 0; same && i < 8; ++i) same = m.key[i] == key[i];
        if (same) { r = m.plan; return m.ok; }
    }
    RowsPlan fresh{};
    const bool ok = rows_plan_search(N, Cin, H, W, Cout, fresh, taps, stride2, nconv);
    Memo& m = memo[next];                  // (claimed after the search: the search itself looks up the one-by-one plan)
    next = (next + 1) & 15;
    for (int i = 0; i < 8; ++i) m.key[i] = key[i];
    m.plan = fresh; m.ok = ok; m.used = true;
    r = fresh;
    return ok;
}

inline bool rows_plan_search(int N, int Cin, int H, int W, int Cout, RowsPlan& r, int taps, bool stride2, int nconv) {
    // (H, W: the map the walk runs over = dY's; with stride2 the input is 2H x 2W)
    if (N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || (W & 7) || (H & ((taps == 1 || stride2) ? 1 : 7))) return false;
    if ((uint64_t)N * Cin * H * W * (stride2 ? 16ull : 4ull) >= 0x40000000ull || (uint64_t)N * Cout * H * W * 4ull >= 0x40000000ull) return false;
    const int tile = taps == 9 ? 32 : 64;                                    // the 1x1 kernel owns 64 x 64 tiles
    const long long tiles = (long long)((Cout + tile - 1) / tile) * ((Cin + tile - 1) / tile);
    r.F = (W % 16 == 0 && !stride2) ? 8 : 4;              // floats per lane and row: strips of 16 / 8 pixels
    r.strips = W / (2 * r.F);
    const int rb = (taps == 1 || stride2) ? 2 : r.F == 4 ? 8 : 4;             // rows per loop body
    const long long sr = (long long)N * H * r.strips;                       // strip-rows in all
    long long splits = MCQ_WGROWS_WAVES / tiles;
    if (splits < 4) splits = 4;
    // (nconv: convolutions sharing the launch.  On small maps the wave budget is the launch's, not each convolution's -- sixteen
    //  16x16 problems planned one by one were 8 192 waves of four strip-rows each and 75 MB of partial sums for 9.7 GFLOP; one
    //  round of 2 048 resident waves of sixteen rows does the same work.  Large maps keep the many-round plan: with hundreds
    //  of rows per wave a launch of slightly more than 2 048 equal waves would end in a nearly empty second round.)
    if (nconv > 1) {
        long long shared = MCQ_WGROWS_WAVES / (tiles * nconv);
        if (shared < 4) shared = 4;
        if (sr / shared <= 64) splits = shared;
    }
    long long by_work = sr * r.F / (4 * MCQ_WGROWS_MIN_ROWS) / (taps == 1 ? 2 : 1);      // (a 1x1 row is 32 MFMAs, not 72)
    if (by_work < 1) by_work = 1;
    if (splits > by_work) splits = by_work;
    long long chunks = (splits + (long long)N * r.strips - 1) / ((long long)N * r.strips);
    const long long hb = H / rb;                                             // row ranges are whole multiples of the body
    if (chunks > hb) chunks = hb;
    r.rpc = (int)((hb + chunks - 1) / chunks) * rb;
    r.row_chunks = (H + r.rpc - 1) / r.rpc;
    r.units = N * r.strips * r.row_chunks;
    r.ups = (int)((r.units + splits - 1) / splits);
    r.splits = (r.units + r.ups - 1) / r.ups;
    r.groups = (r.splits + 3) / 4;
    r.gmax = r.groups;
    if (taps == 9 && !stride2 && sr / MCQ_WGROWS_MIN_ROWS >= 4) {
        // The launch is then planned as a whole.  All its waves are equal; the matrix pipe of a SIMD is kept busy by ONE such
        // wave (two per SIMD take twice as long over the same rows: "1 instead of 2 waves per SIMD changes nothing"),
        // so a launch lasts  ceil(waves / 1024 SIMDs) x (rows per wave + fixed)  with the fixed part (first loads, LDS tree,
        // 36 KB of partial sums per workgroup and their share of the reduce pass) worth about six rows.  Pick the cut (row
        // ranges per column x columns per wave) that minimises it -- twelve 64x64 convolutions planned one by one were 24 576
        // waves of 16 rows (998 us, 226 MB of partials); 3 072 waves of 128 rows do the same work.  Never more groups per
        // convolution than the one-by-one plan above (gmax: what the workspace query reports).
        if (nconv > 1) {
            RowsPlan one;
            (void)rows_plan(N, Cin, H, W, Cout, one);
            r.gmax = one.gmax;
        }
        const long long cols = (long long)N * r.strips, fixed = r.F == 8 ? 6 : 12, simds = MCQ_WGROWS_WAVES / 2;
        long long best = -1; int best_chunks = 0, best_ups = 0;
        for (long long ch = 1; ch <= hb; ++ch) {
            const long long rpc = (hb + ch - 1) / ch * rb, rc = (H + rpc - 1) / rpc, units = cols * rc;
            if (ch > 1 && rpc == (hb + ch - 2) / (ch - 1) * rb) continue;          // (same row ranges as the previous count)
            for (long long ups = 1; ups <= units; ++ups) {
                const long long sp = (units + ups - 1) / ups;
                if (sp < 4) break;
                if ((sp + 3) / 4 > r.gmax) continue;
                const long long waves = tiles * nconv * ((sp + 3) / 4) * 4;
                const long long cost = (waves + simds - 1) / simds * (ups * rpc + fixed);
                if (best < 0 || cost < best) { best = cost; best_chunks = (int)ch; best_ups = (int)ups; }
            }
        }
        if (best >= 0) {
            r.rpc = (int)((hb + best_chunks - 1) / best_chunks) * rb;
            r.row_chunks = (H + r.rpc - 1) / r.rpc;
            r.units = N * r.strips * r.row_chunks;
            r.ups = best_ups;
            r.splits = (r.units + r.ups - 1) / r.ups;
        }
    }
    r.groups = (r.splits + 3) / 4;
    return true;
}

__device__ __forceinline__ f32x4v rows_ld4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

// F = floats a lane reads per row and operand (4 or 8): the strip is 2 F pixels wide and a row is F k-steps = 9 F MFMAs.
// Every lane streams its OWN channel row, so a wave-wide load touches 64 different cache lines whatever F is; what F buys
// is the share of each 128-byte line that is used before it leaves L1 (F = 4: 32 of 128 bytes -- the L2 -> L1 fill path of
// a CU then runs at ~90 % and the matrix pipe at 75 %; F = 8 halves that traffic).  F = 4 remains for maps 8 pixels wide.
// Rings (slot = row mod ring size, all indices compile-time constants: the body covers RB rows):
//   F = 4:  dY 4 slots / 3 rows ahead, X 8 slots / 5 rows ahead        F = 8:  dY 2 slots / 1 row ahead, X 4 slots / 2 rows ahead
// (a row of F = 8 is 72 MFMAs = 4608 cycles, twice that with the SIMD's other wave: one row of lead is ~4 us).
template <bool BIAS, int F>
__global__ __launch_bounds__(256, 2) void conv_wgrad_rows_kernel(WgRowsK p) {
    constexpr int RA = F == 4 ? 4 : 2, LA = RA - 1;           // dY ring / look-ahead in rows
    constexpr int RB = F == 4 ? 8 : 4, LB = F == 4 ? 5 : 2;   // X ring / look-ahead
    __shared__ float red[2][80][64];                          // LDS tree of the workgroup's four waves, five taps at a time
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int kh = lane >> 5, j = lane & 31;
    // blockIdx.x = conv * groups + group: several convolutions of one shape share a launch (the 16x16 ... 8x8 levels of the
    // training step are launch-bound: one 8x8 conv alone gives 128 waves 8 rows each)
    const int conv = (int)blockIdx.x / p.groups, group = (int)blockIdx.x - conv * p.groups;
    const int split = group * 4 + wave;
    const int ci_base = blockIdx.y * 32, co_base = blockIdx.z * 32;
    const float* xp = p.x[0];
    const float* dyp = p.dy[0];
#pragma unroll
    for (int c = 1; c < ROWS_MAX_CONVS; ++c)                  // (select chain: kernel-argument arrays cannot be indexed dynamically without scratch)
        if (c == conv) { xp = p.x[c]; dyp = p.dy[c]; }
    const __amdgpu_buffer_rsrc_t rx = mcq_make_rsrc(xp, (uint32_t)((size_t)p.N * p.Cin * p.H * p.W * 4));
    const __amdgpu_buffer_rsrc_t rd = mcq_make_rsrc(dyp, (uint32_t)((size_t)p.N * p.Cout * p.H * p.W * 4));
    const unsigned rowb = (unsigned)p.W * 4u;
    const bool co_ok = co_base + j < p.Cout, ci_ok = ci_base + j < p.Cin;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    float bsum = 0.0f;

    // The walks of a workgroup are dealt to its four waves round-robin, and consecutive walks are adjacent strips of the same
    // rows (neighbouring pieces of the same cache lines at about the same time).
    for (int i = 0; i < p.ups; ++i) {                         // wave-uniform walk list
        const int u = (group * p.ups + i) * 4 + wave;
        if (u >= p.units) break;
        const int sx = u % p.strips;
        const int t0 = u / p.strips;
        const int chunk = t0 % p.row_chunks, n = t0 / p.row_chunks;
        const int y0 = chunk * p.rpc;
        int y1 = y0 + p.rpc;
        if (y1 > p.H) y1 = p.H;
        const int xl = sx * (2 * F) + F * kh;                 // this lane's first column
        const unsigned vA = co_ok ? (unsigned)(((n * p.Cout + co_base + j) * p.H * p.W + xl) * 4) : MCQ_OOB;
        const unsigned vB = ci_ok ? (unsigned)(((n * p.Cin + ci_base + j) * p.H * p.W + xl) * 4) : MCQ_OOB;
        const unsigned vBl = (ci_ok && xl > 0) ? vB - 4u : MCQ_OOB;
        const unsigned vBr = (ci_ok && xl + F < p.W) ? vB + 4u * F : MCQ_OOB;

        // Row ranges are whole multiples of RB rows (rows_plan), so the walk has no remainder and no branch.
        float A[RA][F];
        float Bw[RB][F + 2];
        auto loadA = [&](int slot, int y) {
#pragma unroll
            for (int v = 0; v < F / 4; ++v) {
                const f32x4v m = rows_ld4(rd, vA + 16u * v, (unsigned)y * rowb);
                A[slot][4 * v + 0] = m[0]; A[slot][4 * v + 1] = m[1]; A[slot][4 * v + 2] = m[2]; A[slot][4 * v + 3] = m[3];
            }
        };
        auto loadB = [&](int slot, int r) {
            const unsigned so = (r >= 0 && r < p.H) ? (unsigned)r * rowb : ROWS_OOB_S;      // scalar select: rows outside the image read 0
            Bw[slot][0] = mcq_buffer_load_s(rx, vBl, so);
#pragma unroll
            for (int v = 0; v < F / 4; ++v) {
                const f32x4v m = rows_ld4(rx, vB + 16u * v, so);
                Bw[slot][4 * v + 1] = m[0]; Bw[slot][4 * v + 2] = m[1]; Bw[slot][4 * v + 3] = m[2]; Bw[slot][4 * v + 4] = m[3];
            }
            Bw[slot][F + 1] = mcq_buffer_load_s(rx, vBr, so);
        };
#pragma unroll
        for (int s = 0; s < LA; ++s) loadA(s, y0 + s);
        loadB(RB - 1, y0 - 1);
#pragma unroll
        for (int s = 0; s < LB; ++s) loadB(s, y0 + s);

        for (int yb = y0; yb < y1; yb += RB) {
#pragma unroll
            for (int uu = 0; uu < RB; ++uu) {
                const int y = yb + uu;
                const int sa = uu % RA, sprev = (uu + RB - 1) % RB, snext = (uu + 1) % RB;
                loadB((uu + LB) % RB, y + LB);                // (a slot no row of y - 1 .. y + 1 lives in)
                loadA((uu + LA) % RA, y + LA);                // (the slot of row y - 1)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < F; ++q)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
                        acc[dx] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[sa][q], Bw[sprev][q + dx], acc[dx], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < F; ++q)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
                        acc[3 + dx] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[sa][q], Bw[uu][q + dx], acc[3 + dx], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < F; ++q)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
                        acc[6 + dx] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[sa][q], Bw[snext][q + dx], acc[6 + dx], 0, 0, 0);
                if (BIAS) {
#pragma unroll
                    for (int q = 0; q < F; q += 4) bsum = bsum + ((A[sa][q] + A[sa][q + 1]) + (A[sa][q + 2] + A[sa][q + 3]));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- the four waves of the workgroup meet in LDS: (w0 + w1) + (w2 + w3), five taps at a time --------------------
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int tb = half * 5, tn = half ? 4 : 5;
        if (wave & 1) {
#pragma unroll
            for (int t = 0; t < 5; ++t)
                if (t < tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[wave >> 1][t * 16 + r][lane] = acc[tb + t][r];
        }
        __syncthreads();
        if (!(wave & 1)) {
#pragma unroll
            for (int t = 0; t < 5; ++t)
                if (t < tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tb + t][r] = acc[tb + t][r] + red[wave >> 1][t * 16 + r][lane];
        }
        __syncthreads();
        if (wave == 2) {
#pragma unroll
            for (int t = 0; t < 5; ++t)
                if (t < tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[0][t * 16 + r][lane] = acc[tb + t][r];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int t = 0; t < 5; ++t)
                if (t < tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tb + t][r] = acc[tb + t][r] + red[0][t * 16 + r][lane];
        }
        __syncthreads();
    }
    if (wave == 0) {
        // part[group][tap][co][ci]: 32 lanes = 32 consecutive ci = one 128-byte line
        float* out = p.part + (size_t)blockIdx.x * 9 * p.Cout * p.Cin;        // [conv][group] = blockIdx.x
        const int ci = ci_base + j;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co_base + mcq_drow(r, kh);
                if (co < p.Cout && ci < p.Cin) out[((size_t)t * p.Cout + co) * p.Cin + ci] = acc[t][r];
            }
    }
    if (BIAS && blockIdx.y == 0) {                            // the ci-tile-0 waves leave their bias partials (tiny)
        const float s = bsum + __shfl_xor(bsum, 32);          // the two pixel halves
        if (kh == 0 && co_ok) p.bias_part[((size_t)conv * p.groups * 4 + split) * p.Cout + co_base + j] = s;
    }
    // (round 4, measured and removed: a last-arriver tail -- the workgroups of a (convolution, tile) count themselves in on a
    //  self-resetting counter and the last one adds the groups' partials, so that the second launch disappears where the groups
    //  are few.  Correct and deterministic, and the captured training step went from 22.27 to 23.66 ms (G <= 8; 23.26 at G <= 4,
    //  23.90 at G <= 16): the agent-scope release in EVERY workgroup writes the XCD's L2 back and the adder's acquire
    //  invalidates it, which costs the following launches far more than the 54 reduce launches of 5-28 us it saves.)
}

struct WgRowsReduceK {
    const float* part; const float* bias_part; float* dw[ROWS_MAX_CONVS]; float* dbias[ROWS_MAX_CONVS];
    int nconv, groups, Cout, Cin, taps;
};

// dW[conv][co][ci][tap] = sum_g part[conv][g][tap][co][ci]; db[conv][co] = sum_s bias_part[conv][s][co]   (fixed order)
__global__ __launch_bounds__(256) void wgrad_rows_reduce_kernel(WgRowsReduceK p) {
    // a workgroup = 64 outputs x 4 slices of the group range (eight independent loads in flight per thread; the slice sums
    // meet in LDS and are added in slice order)
    __shared__ float red[4][64];
    const int conv = blockIdx.y;
    float* dw = p.dw[0];
    float* dbias = p.dbias[0];
#pragma unroll
    for (int c = 1; c < ROWS_MAX_CONVS; ++c)
        if (c == conv) { dw = p.dw[c]; dbias = p.dbias[c]; }
    const int Cout = p.Cout, Cin = p.Cin;
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * 64 + lane;                    // index into [tap][co][ci], then [co] of the bias
    const size_t per = (size_t)p.taps * Cout * Cin;
    const bool is_bias = i >= per;
    const size_t co_b = i - per;
    const bool live = is_bias ? (dbias != nullptr && co_b < (size_t)Cout) : true;
    const int count = is_bias ? p.groups * 4 : p.groups;
    const int per_slice = (count + 3) / 4;
    const int s0 = slice * per_slice, s1 = s0 + per_slice < count ? s0 + per_slice : count;
    const float* src = is_bias ? p.bias_part + (size_t)conv * p.groups * 4 * Cout + co_b : p.part + (size_t)conv * p.groups * per + i;
    const size_t stride = is_bias ? (size_t)Cout : per;
    float s = 0.0f;
    if (live) {
        int g = s0;
        for (; g + 8 <= s1; g += 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = src[(size_t)(g + k) * stride];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
        for (; g < s1; ++g) s += src[(size_t)g * stride];
    }
    red[slice][lane] = s;
    __syncthreads();
    if (slice != 0 || !live) return;
    s = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
    if (is_bias) { dbias[co_b] = s; return; }
    const int ci = (int)(i % Cin);
    const size_t r = i / Cin;
    const int co = (int)(r % Cout);
    const int tap = (int)(r / Cout);
    dw[((size_t)co * Cin + ci) * p.taps + tap] = s;
}

// ---- the reduce passes of a whole backward pass in a few launches (round 5) ------------------------------------------------------
// A captured training step held 56 reduce launches, 44 of them 5-9 us of latency for microseconds of traffic.  While deferral is
// on (mcq_wgrad_defer; mcquic_amd.autograd.backward switches it on around a backward pass it owns) the entry points below record
// their reduce pass per convolution instead of launching it, and mcq_wgrad_flush launches all recorded passes, up to
// REDUCE_BATCH per launch (blockIdx.y = the convolution).  Same sums in the same order as the
// one-by-one kernel.  The caller keeps every workspace alive until the flush and reads no weight gradient before it.
constexpr int REDUCE_BATCH = 80;
struct ReduceJob { const float* part; const float* bias_part; float* dw; float* dbias; int groups, Cout, Cin, taps; };
struct ReduceBatch { ReduceJob job[REDUCE_BATCH]; };

// grid (blocks of the largest job, jobs): a job's surplus blocks leave at once.  (The first form put the jobs' blocks back to back
// on blockIdx.x and searched a prefix table: 72 dependent scalar loads per workgroup, 3 us each -- four times the one-by-one passes.)
__global__ __launch_bounds__(256) void wgrad_rows_reduce_batch_kernel(ReduceBatch b) {
    __shared__ float red[4][64];
    const ReduceJob q = b.job[blockIdx.y];                              // (uniform index into the kernel-argument table)
    const int Cout = q.Cout, Cin = q.Cin;
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const size_t per = (size_t)q.taps * Cout * Cin;
    if ((size_t)blockIdx.x * 64 >= per + (q.dbias ? (size_t)Cout : 0)) return;      // (workgroup-uniform)
    const size_t i = (size_t)blockIdx.x * 64 + lane;
    const bool is_bias = i >= per;
    const size_t co_b = i - per;
    const bool live = is_bias ? (q.dbias != nullptr && co_b < (size_t)Cout) : true;
    const int count = is_bias ? q.groups * 4 : q.groups;
    const int per_slice = (count + 3) / 4;
    const int s0 = slice * per_slice, s1 = s0 + per_slice < count ? s0 + per_slice : count;
    const float* src = is_bias ? q.bias_part + co_b : q.part + i;
    const size_t stride = is_bias ? (size_t)Cout : per;
    float s = 0.0f;
    if (live) {
        int g = s0;
        for (; g + 8 <= s1; g += 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = src[(size_t)(g + k) * stride];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
        for (; g < s1; ++g) s += src[(size_t)g * stride];
    }
    red[slice][lane] = s;
    __syncthreads();
    if (slice != 0 || !live) return;
    s = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
    if (is_bias) { q.dbias[co_b] = s; return; }
    const int ci = (int)(i % Cin);
    const size_t r = i / Cin;
    const int co = (int)(r % Cout);
    const int tap = (int)(r / Cout);
    q.dw[((size_t)co * Cin + ci) * q.taps + tap] = s;
}

std::mutex g_defer_mu;
bool g_defer = false;
// recorded reduce passes, per device (a job is only ever flushed onto a stream of the device its buffers live on)
std::unordered_map<int, std::vector<ReduceJob>> g_jobs;
inline int current_device() { int d = 0; (void)hipGetDevice(&d); return d; }

// the second pass of a weight-gradient launch: now, or recorded per convolution for mcq_wgrad_flush
inline void reduce_pass(const WgRowsReduceK& q, unsigned gx, unsigned nconv, hipStream_t s) {
    {
        std::lock_guard<std::mutex> lock(g_defer_mu);
        if (g_defer) {
            const size_t per = (size_t)q.taps * q.Cout * q.Cin;
            std::vector<ReduceJob>& mine = g_jobs[current_device()];
            for (unsigned c = 0; c < nconv; ++c)
                mine.push_back(ReduceJob{q.part + (size_t)c * q.groups * per, q.bias_part ? q.bias_part + (size_t)c * q.groups * 4 * q.Cout : nullptr,
                                           q.dw[c], q.dbias[c], q.groups, q.Cout, q.Cin, q.taps});
            return;
        }
    }
    hipLaunchKernelGGL(wgrad_rows_reduce_kernel, dim3(gx, nconv), dim3(256), 0, s, q);
}

// ---- 3x3 stride-2 convolutions (ResidualBlockWithStride's two convs, the 3-channel stem) --------------------------------
//   dW[co][ci][dy][dx] = sum dY[n][co][yo][xo] * X[n][ci][2 yo + dy - 1][2 xo + dx - 1]
// The walk runs over dY's rows; a lane reads 4 consecutive dY pixels and, per input row, the 9 input columns 2 x0 - 1 .. 2 x0 + 7
// they touch (two 16-byte loads + the left neighbour): tap (dy, dx) of k-step q is window register 2 q + dx of input row
// 2 yo + dy - 1.  Output row yo + 1 re-uses input row 2 yo + 1, so two new input rows arrive per output row, in a ring of four
// (slot = row mod 4; the slot of row 2 yo - 1 is re-filled right after the dy = 0 taps).  Only the top border is padding
// (even H, W: row -1 and column -1), handled like in the stride-1 walk.  Same tile, LDS tree, partial layout and reduce pass.
template <bool BIAS>
__global__ __launch_bounds__(256, 2) void conv_wgrad_rows_s2_kernel(WgRowsK p) {
    __shared__ float red[2][80][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int kh = lane >> 5, j = lane & 31;
    const int group = blockIdx.x;
    const int split = group * 4 + wave;
    const int ci_base = blockIdx.y * 32, co_base = blockIdx.z * 32;
    const int Hi = 2 * p.H, Wi = 2 * p.W;                      // the input map
    const __amdgpu_buffer_rsrc_t rx = mcq_make_rsrc(p.x[0], (uint32_t)((size_t)p.N * p.Cin * Hi * Wi * 4));
    const __amdgpu_buffer_rsrc_t rd = mcq_make_rsrc(p.dy[0], (uint32_t)((size_t)p.N * p.Cout * p.H * p.W * 4));
    const unsigned rowd = (unsigned)p.W * 4u, rowx = (unsigned)Wi * 4u;
    const bool co_ok = co_base + j < p.Cout, ci_ok = ci_base + j < p.Cin;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    float bsum = 0.0f;
    for (int i = 0; i < p.ups; ++i) {
        const int u = (group * p.ups + i) * 4 + wave;
        if (u >= p.units) break;
        const int sx = u % p.strips;
        const int t0 = u / p.strips;
        const int chunk = t0 % p.row_chunks, n = t0 / p.row_chunks;
        const int y0 = chunk * p.rpc;
        int y1 = y0 + p.rpc;
        if (y1 > p.H) y1 = p.H;
        const int xo0 = sx * 8 + 4 * kh;                      // this lane's first output column
        const unsigned vA = co_ok ? (unsigned)(((n * p.Cout + co_base + j) * p.H * p.W + xo0) * 4) : MCQ_OOB;
        const unsigned vB = ci_ok ? (unsigned)(((n * p.Cin + ci_base + j) * Hi * Wi + 2 * xo0) * 4) : MCQ_OOB;
        const unsigned vBl = (ci_ok && xo0 > 0) ? vB - 4u : MCQ_OOB;
        float A[2][4];
        float Bw[4][9];
        auto loadA = [&](int slot, int y) {
            const f32x4v m = rows_ld4(rd, vA, (unsigned)y * rowd);
            A[slot][0] = m[0]; A[slot][1] = m[1]; A[slot][2] = m[2]; A[slot][3] = m[3];
        };
        auto loadB = [&](int slot, int r) {
            const unsigned so = (r >= 0 && r < Hi) ? (unsigned)r * rowx : ROWS_OOB_S;
            Bw[slot][0] = mcq_buffer_load_s(rx, vBl, so);
            const f32x4v m0 = rows_ld4(rx, vB, so), m1 = rows_ld4(rx, vB + 16u, so);
            Bw[slot][1] = m0[0]; Bw[slot][2] = m0[1]; Bw[slot][3] = m0[2]; Bw[slot][4] = m0[3];
            Bw[slot][5] = m1[0]; Bw[slot][6] = m1[1]; Bw[slot][7] = m1[2]; Bw[slot][8] = m1[3];
        };
        // y0 is even (row ranges are whole multiples of 2 rows), so input row r lives in slot r mod 4 with 2 y0 = 0 mod 4 ... only
        // if y0 is even in units of 2: use slots relative to the walk instead: row 2 (y0 + t) + d - 1 -> slot (2 t + d + 3) mod 4
        loadA(0, y0);
        loadB(3, 2 * y0 - 1);
        loadB(0, 2 * y0);
        loadB(1, 2 * y0 + 1);
        for (int yb = y0; yb < y1; yb += 2) {
#pragma unroll
            for (int uu = 0; uu < 2; ++uu) {
                const int yo = yb + uu;
                const int s0 = uu == 0 ? 3 : 1, s1 = uu == 0 ? 0 : 2, s2 = uu == 0 ? 1 : 3;   // rows 2 yo - 1, 2 yo, 2 yo + 1
                const int snew = uu == 0 ? 2 : 0;                                               // free: takes row 2 yo + 2
                loadB(snew, 2 * yo + 2);
                loadA(uu ^ 1, yo + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
                        acc[dx] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[uu][q], Bw[s0][2 * q + dx], acc[dx], 0, 0, 0);
                loadB(s0, 2 * yo + 3);                        // (the slot of row 2 yo - 1, done with)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
                        acc[3 + dx] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[uu][q], Bw[s1][2 * q + dx], acc[3 + dx], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
                        acc[6 + dx] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[uu][q], Bw[s2][2 * q + dx], acc[6 + dx], 0, 0, 0);
                if (BIAS) bsum = bsum + ((A[uu][0] + A[uu][1]) + (A[uu][2] + A[uu][3]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int tb = half * 5, tn = half ? 4 : 5;
        if (wave & 1) {
#pragma unroll
            for (int t = 0; t < 5; ++t)
                if (t < tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[wave >> 1][t * 16 + r][lane] = acc[tb + t][r];
        }
        __syncthreads();
        if (!(wave & 1)) {
#pragma unroll
            for (int t = 0; t < 5; ++t)
                if (t < tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tb + t][r] = acc[tb + t][r] + red[wave >> 1][t * 16 + r][lane];
        }
        __syncthreads();
        if (wave == 2) {
#pragma unroll
            for (int t = 0; t < 5; ++t)
                if (t < tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[0][t * 16 + r][lane] = acc[tb + t][r];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int t = 0; t < 5; ++t)
                if (t < tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tb + t][r] = acc[tb + t][r] + red[0][t * 16 + r][lane];
        }
        __syncthreads();
    }
    if (wave == 0) {
        float* out = p.part + (size_t)group * 9 * p.Cout * p.Cin;
        const int ci = ci_base + j;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co_base + mcq_drow(r, kh);
                if (co < p.Cout && ci < p.Cin) out[((size_t)t * p.Cout + co) * p.Cin + ci] = acc[t][r];
            }
    }
    if (BIAS && blockIdx.y == 0) {
        const float sv = bsum + __shfl_xor(bsum, 32);
        if (kh == 0 && co_ok) p.bias_part[(size_t)split * p.Cout + co_base + j] = sv;
    }
}

// ---- 1x1 convolutions (the AttentionBlock gate conv, GDN's gamma) ------------------------------------------------------
//   dW[co][ci] = sum_{n, y, x} dY[n][co][y][x] * X[n][ci][y][x]            (SQ: X squared -- gamma multiplies x^2, gdn.py:75)
// The same walk without a window: a wave owns a 64 co x 64 ci tile (4 accumulator tiles), reads F = 8 (4) consecutive pixels
// per lane and row for its two dY and two X bands, 4 MFMAs per k-step, next row's operands requested one row ahead.
template <bool BIAS, bool SQ, int F>
__global__ __launch_bounds__(256, 2) void conv_wgrad_rows1_kernel(WgRowsK p) {
    __shared__ float red[2][64][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int kh = lane >> 5, j = lane & 31;
    const int group = blockIdx.x;
    const int split = group * 4 + wave;
    const int ci_base = blockIdx.y * 64, co_base = blockIdx.z * 64;
    const __amdgpu_buffer_rsrc_t rx = mcq_make_rsrc(p.x[0], (uint32_t)((size_t)p.N * p.Cin * p.H * p.W * 4));
    const __amdgpu_buffer_rsrc_t rd = mcq_make_rsrc(p.dy[0], (uint32_t)((size_t)p.N * p.Cout * p.H * p.W * 4));
    const unsigned rowb = (unsigned)p.W * 4u;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    float bsum[2] = {0.0f, 0.0f};
    for (int i = 0; i < p.ups; ++i) {
        const int u = (group * p.ups + i) * 4 + wave;
        if (u >= p.units) break;
        const int sx = u % p.strips;
        const int t0 = u / p.strips;
        const int chunk = t0 % p.row_chunks, n = t0 / p.row_chunks;
        const int y0 = chunk * p.rpc;
        int y1 = y0 + p.rpc;
        if (y1 > p.H) y1 = p.H;
        const int xl = sx * (2 * F) + F * kh;
        unsigned vA[2], vB[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            vA[t] = co_base + 32 * t + j < p.Cout ? (unsigned)(((n * p.Cout + co_base + 32 * t + j) * p.H * p.W + xl) * 4) : MCQ_OOB;
            vB[t] = ci_base + 32 * t + j < p.Cin ? (unsigned)(((n * p.Cin + ci_base + 32 * t + j) * p.H * p.W + xl) * 4) : MCQ_OOB;
        }
        float A[2][2][F], B[2][2][F];                         // [slot][band][pixel]
        auto load = [&](int slot, int y) {
            const unsigned so = (unsigned)y * rowb;           // (rows past the range are requested but never used)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int v = 0; v < F / 4; ++v) {
                    const f32x4v a = rows_ld4(rd, vA[t] + 16u * v, so);
                    const f32x4v b = rows_ld4(rx, vB[t] + 16u * v, so);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { A[slot][t][4 * v + e] = a[e]; B[slot][t][4 * v + e] = b[e]; }
                }
        };
        load(0, y0);
        for (int yb = y0; yb < y1; yb += 2) {
#pragma unroll
            for (int uu = 0; uu < 2; ++uu) {
                load(uu ^ 1, yb + uu + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < F; ++q) {
                    float b0 = B[uu][0][q], b1 = B[uu][1][q];
                    if (SQ) { b0 = b0 * b0; b1 = b1 * b1; }
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[uu][0][q], b0, acc[0][0], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[uu][1][q], b0, acc[1][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[uu][0][q], b1, acc[0][1], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[uu][1][q], b1, acc[1][1], 0, 0, 0);
                    if (BIAS) { bsum[0] = bsum[0] + A[uu][0][q]; bsum[1] = bsum[1] + A[uu][1][q]; }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // LDS tree of the four waves, one accumulator tile pair at a time: (w0 + w1) + (w2 + w3)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        if (wave & 1) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[wave >> 1][a * 16 + r][lane] = acc[a][b][r];
        }
        __syncthreads();
        if (!(wave & 1)) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = acc[a][b][r] + red[wave >> 1][a * 16 + r][lane];
        }
        __syncthreads();
        if (wave == 2) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[0][a * 16 + r][lane] = acc[a][b][r];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = acc[a][b][r] + red[0][a * 16 + r][lane];
        }
        __syncthreads();
    }
    if (wave == 0) {
        float* out = p.part + (size_t)group * p.Cout * p.Cin;                       // part[group][co][ci]
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co_base + 32 * a + mcq_drow(r, kh), ci = ci_base + 32 * b + j;
                    if (co < p.Cout && ci < p.Cin) out[(size_t)co * p.Cin + ci] = acc[a][b][r];
                }
    }
    if (BIAS && blockIdx.y == 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float sv = bsum[t] + __shfl_xor(bsum[t], 32);
            const int co = co_base + 32 * t + j;
            if (kh == 0 && co < p.Cout) p.bias_part[(size_t)split * p.Cout + co] = sv;
        }
    }
}

}  // namespace

namespace {

// ---- maps too small for a strip walk (the 4x4 level of a 256x256 crop: 16 pixels per image) ---------------------------
// One thread per (co, ci) pair of a 16 x 16 tile, its nine taps in registers; the tile's dY planes and zero-bordered X
// planes sit in LDS (8 images at a time), so the border needs no test.  19 MFLOP per conv: the point is ONE launch instead
// of transposes + GEMM + reduction, not the arithmetic rate.  Deterministic (fixed loop order).
constexpr int TINY_MAX_HW = 64, TINY_IMGS = 8;

struct WgTinyK {
    const float* x[ROWS_MAX_CONVS]; const float* dy[ROWS_MAX_CONVS]; float* dw[ROWS_MAX_CONVS]; float* dbias[ROWS_MAX_CONVS];
    int N, Cin, Cout, H, W;
};

__global__ __launch_bounds__(256) void conv_wgrad_tiny_kernel(WgTinyK p) {
    extern __shared__ float lds[];
    const int conv = blockIdx.z;
    const float* xp = p.x[0];
    const float* dyp = p.dy[0];
    float* dw = p.dw[0];
    float* dbias = p.dbias[0];
#pragma unroll
    for (int c = 1; c < ROWS_MAX_CONVS; ++c)
        if (c == conv) { xp = p.x[c]; dyp = p.dy[c]; dw = p.dw[c]; dbias = p.dbias[c]; }
    const int HW = p.H * p.W, PW = p.W + 2;
    const int plane = ((p.H + 2) * PW) | 1;                   // zero-bordered X plane, odd pitch (no LDS bank conflicts)
    float* dyl = lds;                                         // [img][16 co][HW]
    float* xl = lds + TINY_IMGS * 16 * HW;                    // [img][16 ci][plane]
    const int ci_l = threadIdx.x & 15, co_l = threadIdx.x >> 4;
    const int ci0 = blockIdx.x * 16, co0 = blockIdx.y * 16;
    float acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = 0.0f;
    float bsum = 0.0f;
    for (int n0 = 0; n0 < p.N; n0 += TINY_IMGS) {
        const int nn = p.N - n0 < TINY_IMGS ? p.N - n0 : TINY_IMGS;
        __syncthreads();
        for (int i = threadIdx.x; i < nn * 16 * plane; i += 256) xl[i] = 0.0f;
        __syncthreads();
        for (int i = threadIdx.x; i < nn * 16 * HW; i += 256) {
            const int px = i % HW, c = (i / HW) & 15, n = i / (HW * 16);
            dyl[i] = co0 + c < p.Cout ? dyp[((size_t)(n0 + n) * p.Cout + co0 + c) * HW + px] : 0.0f;
            if (ci0 + c < p.Cin)
                xl[(n * 16 + c) * plane + (px / p.W + 1) * PW + (px % p.W) + 1] = xp[((size_t)(n0 + n) * p.Cin + ci0 + c) * HW + px];
        }
        __syncthreads();
        for (int n = 0; n < nn; ++n) {
            const float* dr = dyl + (n * 16 + co_l) * HW;
            const float* xr = xl + (n * 16 + ci_l) * plane;
            for (int y = 0; y < p.H; ++y)
                for (int x = 0; x < p.W; ++x) {
                    const float d = dr[y * p.W + x];
                    bsum += d;
                    const float* w0 = xr + y * PW + x;
#pragma unroll
                    for (int t = 0; t < 9; ++t) acc[t] = __builtin_fmaf(d, w0[(t / 3) * PW + (t % 3)], acc[t]);
                }
        }
    }
    const int co = co0 + co_l, ci = ci0 + ci_l;
    if (co < p.Cout && ci < p.Cin) {
#pragma unroll
        for (int t = 0; t < 9; ++t) dw[((size_t)co * p.Cin + ci) * 9 + t] = acc[t];
    }
    if (dbias && blockIdx.x == 0 && ci_l == 0 && co < p.Cout) dbias[co] = bsum;
}

inline size_t tiny_lds_bytes(int H, int W) {
    return (size_t)TINY_IMGS * 16 * (H * W + (((H + 2) * (W + 2)) | 1)) * sizeof(float);
}

inline bool tiny_shape(int N, int Cin, int H, int W, int Cout) {
    return N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && H * W <= TINY_MAX_HW && tiny_lds_bytes(H, W) <= 65536 && (uint64_t)N * (Cin > Cout ? Cin : Cout) * H * W < (1ull << 28);
}
}  // namespace

#include "wgrad_t16.h"

extern "C" int32_t mcq_conv2d_wgrad_nchw_max_group(void) { return ROWS_MAX_CONVS; }

extern "C" size_t mcq_conv2d_wgrad_nchw_workspace_floats(int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t Cout) {
    if (wgt16_shape(N, Cin, H, W, Cout)) return 1;                                            // (wgrad_t16.h: one pass, no workspace)
    RowsPlan r;
    if (!rows_plan(N, Cin, H, W, Cout, r)) return tiny_shape(N, Cin, H, W, Cout) ? 1 : 0;      // (the small-map kernel needs no workspace)
    return (size_t)r.gmax * 9 * Cout * Cin + (size_t)r.gmax * 4 * Cout;
}

extern "C" int mcq_conv2d_wgrad_nchw_group_f32(const float* const* x, const float* const* dy, float* const* dw, float* const* dbias,
                                               int32_t nconv, float* workspace, int32_t N, int32_t Cin, int32_t H, int32_t W,
                                               int32_t Cout, void* stream) {
    if (!x || !dy || !dw || !workspace || nconv < 1 || nconv > ROWS_MAX_CONVS) return MCQ_EINVAL;
    if (wgt16_shape(N, Cin, H, W, Cout)) {                          // few pixels: 16 x 16 tiles, one pass (wgrad_t16.h)
        for (int c = 0; c < nconv; ++c)
            if (!x[c] || !dy[c] || !dw[c]) return MCQ_EINVAL;
        wgt16_launch(x, dy, dw, dbias, nconv, N, Cin, H, W, Cout, 9, false, (hipStream_t)stream);
        return mcq_check_launch();
    }
    RowsPlan r;
    bool planned = rows_plan(N, Cin, H, W, Cout, r, 9, false, nconv);
    if (planned && nconv > 1) {
        // the workspace query sizes by the one-by-one plan's gmax: a grouped plan that would need more groups than that (the
        // shared-splits path rounds upwards) falls back to the one-by-one plan instead of overrunning the workspace
        RowsPlan one;
        if (rows_plan(N, Cin, H, W, Cout, one) && r.groups > one.gmax) r = one;
    }
    if (!planned) {      // (never more groups than the nconv = 1 plan the workspace query assumes)
        if (!tiny_shape(N, Cin, H, W, Cout)) return MCQ_EINVAL;
        WgTinyK t;
        for (int c = 0; c < ROWS_MAX_CONVS; ++c) {
            const int k = c < nconv ? c : 0;
            if (!x[k] || !dy[k] || !dw[k]) return MCQ_EINVAL;
            t.x[c] = x[k]; t.dy[c] = dy[k]; t.dw[c] = dw[k]; t.dbias[c] = dbias ? dbias[k] : nullptr;
        }
        t.N = N; t.Cin = Cin; t.Cout = Cout; t.H = H; t.W = W;
        const dim3 grid((unsigned)((Cin + 15) / 16), (unsigned)((Cout + 15) / 16), (unsigned)nconv);
        hipLaunchKernelGGL(conv_wgrad_tiny_kernel, grid, dim3(256), tiny_lds_bytes(H, W), (hipStream_t)stream, t);
        return mcq_check_launch();
    }
    WgRowsK p{};
    WgRowsReduceK q;
    bool any_bias = false;
    for (int c = 0; c < ROWS_MAX_CONVS; ++c) {
        const int k = c < nconv ? c : 0;
        if (!x[k] || !dy[k] || !dw[k]) return MCQ_EINVAL;
        p.x[c] = x[k]; p.dy[c] = dy[k];
        q.dw[c] = dw[k]; q.dbias[c] = dbias ? dbias[k] : nullptr;
        any_bias = any_bias || q.dbias[c] != nullptr;
    }
    const size_t per_conv = (size_t)r.groups * 9 * Cout * Cin;
    p.part = workspace;
    p.bias_part = any_bias ? workspace + (size_t)nconv * per_conv : nullptr;
    p.nconv = nconv; p.groups = r.groups;
    p.N = N; p.Cin = Cin; p.Cout = Cout; p.H = H; p.W = W;
    p.strips = r.strips; p.row_chunks = r.row_chunks; p.rpc = r.rpc; p.units = r.units; p.splits = r.splits; p.ups = r.ups;
    q.part = workspace; q.bias_part = p.bias_part; q.nconv = nconv; q.groups = r.groups; q.Cout = Cout; q.Cin = Cin; q.taps = 9;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)(r.groups * nconv), (unsigned)((Cin + 31) / 32), (unsigned)((Cout + 31) / 32));
    if (r.F == 8) {
        if (any_bias) hipLaunchKernelGGL((conv_wgrad_rows_kernel<true, 8>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv_wgrad_rows_kernel<false, 8>), grid, dim3(256), 0, s, p);
    } else {
        if (any_bias) hipLaunchKernelGGL((conv_wgrad_rows_kernel<true, 4>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv_wgrad_rows_kernel<false, 4>), grid, dim3(256), 0, s, p);
    }
    const size_t per = (size_t)9 * Cout * Cin + (any_bias ? (size_t)Cout : 0);
    reduce_pass(q, (unsigned)((per + 63) / 64), (unsigned)nconv, s);
    return mcq_check_launch();
}

extern "C" int mcq_conv2d_wgrad_nchw_f32(const float* x, const float* dy, float* dw, float* dbias, float* workspace, int32_t N,
                                         int32_t Cin, int32_t H, int32_t W, int32_t Cout, void* stream) {
    const float* xs[1] = {x};
    const float* dys[1] = {dy};
    float* dws[1] = {dw};
    float* dbs[1] = {dbias};
    return mcq_conv2d_wgrad_nchw_group_f32(xs, dys, dws, dbias ? dbs : nullptr, 1, workspace, N, Cin, H, W, Cout, stream);
}

extern "C" size_t mcq_conv2d_wgrad1x1_nchw_workspace_floats(int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t Cout) {
    if (wgt16_shape(N, Cin, H, W, Cout)) return 1;
    RowsPlan r;
    if ((H & 1) || !rows_plan(N, Cin, H, W, Cout, r, 1)) return 0;
    return (size_t)r.groups * Cout * Cin + (size_t)r.groups * 4 * Cout;
}

extern "C" int mcq_conv2d_wgrad1x1_nchw_f32(const float* x, const float* dy, float* dw, float* dbias, float* workspace, int32_t N,
                                            int32_t Cin, int32_t H, int32_t W, int32_t Cout, int32_t square_x, void* stream) {
    if (!x || !dy || !dw || !workspace) return MCQ_EINVAL;
    if (wgt16_shape(N, Cin, H, W, Cout)) {                          // few pixels: 16 x 16 tiles, one pass (wgrad_t16.h)
        const float* xs[1] = {x};
        const float* dys[1] = {dy};
        float* dws[1] = {dw};
        float* dbs[1] = {dbias};
        wgt16_launch(xs, dys, dws, dbias ? dbs : nullptr, 1, N, Cin, H, W, Cout, 1, square_x != 0, (hipStream_t)stream);
        return mcq_check_launch();
    }
    RowsPlan r;
    if ((H & 1) || !rows_plan(N, Cin, H, W, Cout, r, 1)) return MCQ_EINVAL;
    WgRowsK p{};
    WgRowsReduceK q;
    for (int c = 0; c < ROWS_MAX_CONVS; ++c) { p.x[c] = x; p.dy[c] = dy; q.dw[c] = dw; q.dbias[c] = dbias; }
    p.part = workspace;
    p.bias_part = dbias ? workspace + (size_t)r.groups * Cout * Cin : nullptr;
    p.nconv = 1; p.groups = r.groups;
    p.N = N; p.Cin = Cin; p.Cout = Cout; p.H = H; p.W = W;
    p.strips = r.strips; p.row_chunks = r.row_chunks; p.rpc = r.rpc; p.units = r.units; p.splits = r.splits; p.ups = r.ups;
    q.part = workspace; q.bias_part = p.bias_part; q.nconv = 1; q.groups = r.groups; q.Cout = Cout; q.Cin = Cin; q.taps = 1;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)r.groups, (unsigned)((Cin + 63) / 64), (unsigned)((Cout + 63) / 64));
#define MCQ_LAUNCH_ROWS1(B_, S_, F_) hipLaunchKernelGGL((conv_wgrad_rows1_kernel<B_, S_, F_>), grid, dim3(256), 0, s, p)
    const bool b = dbias != nullptr, sq = square_x != 0;
    if (r.F == 8) {
        if (b) { if (sq) MCQ_LAUNCH_ROWS1(true, true, 8); else MCQ_LAUNCH_ROWS1(true, false, 8); }
        else { if (sq) MCQ_LAUNCH_ROWS1(false, true, 8); else MCQ_LAUNCH_ROWS1(false, false, 8); }
    } else {
        if (b) { if (sq) MCQ_LAUNCH_ROWS1(true, true, 4); else MCQ_LAUNCH_ROWS1(true, false, 4); }
        else { if (sq) MCQ_LAUNCH_ROWS1(false, true, 4); else MCQ_LAUNCH_ROWS1(false, false, 4); }
    }
#undef MCQ_LAUNCH_ROWS1
    const size_t per = (size_t)Cout * Cin + (dbias ? (size_t)Cout : 0);
    reduce_pass(q, (unsigned)((per + 63) / 64), 1u, s);
    return mcq_check_launch();
}

extern "C" size_t mcq_conv2d_wgrad_s2_nchw_workspace_floats(int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t Cout) {
    RowsPlan r;
    if ((H & 1) || (W & 1) || !rows_plan(N, Cin, H / 2, W / 2, Cout, r, 9, true)) return 0;
    return (size_t)r.gmax * 9 * Cout * Cin + (size_t)r.gmax * 4 * Cout;
}

extern "C" int mcq_conv2d_wgrad_s2_nchw_f32(const float* x, const float* dy, float* dw, float* dbias, float* workspace, int32_t N,
                                            int32_t Cin, int32_t H, int32_t W, int32_t Cout, void* stream) {
    if (!x || !dy || !dw || !workspace) return MCQ_EINVAL;
    RowsPlan r;
    if ((H & 1) || (W & 1) || !rows_plan(N, Cin, H / 2, W / 2, Cout, r, 9, true)) return MCQ_EINVAL;
    WgRowsK p{};
    WgRowsReduceK q;
    for (int c = 0; c < ROWS_MAX_CONVS; ++c) { p.x[c] = x; p.dy[c] = dy; q.dw[c] = dw; q.dbias[c] = dbias; }
    p.part = workspace;
    p.bias_part = dbias ? workspace + (size_t)r.groups * 9 * Cout * Cin : nullptr;
    p.nconv = 1; p.groups = r.groups;
    p.N = N; p.Cin = Cin; p.Cout = Cout; p.H = H / 2; p.W = W / 2;          // the walk's map = dY's
    p.strips = r.strips; p.row_chunks = r.row_chunks; p.rpc = r.rpc; p.units = r.units; p.splits = r.splits; p.ups = r.ups;
    q.part = workspace; q.bias_part = p.bias_part; q.nconv = 1; q.groups = r.groups; q.Cout = Cout; q.Cin = Cin; q.taps = 9;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)r.groups, (unsigned)((Cin + 31) / 32), (unsigned)((Cout + 31) / 32));
    if (dbias) hipLaunchKernelGGL(conv_wgrad_rows_s2_kernel<true>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(conv_wgrad_rows_s2_kernel<false>, grid, dim3(256), 0, s, p);
    const size_t per = (size_t)9 * Cout * Cin + (dbias ? (size_t)Cout : 0);
    reduce_pass(q, (unsigned)((per + 63) / 64), 1u, s);
    return mcq_check_launch();
}

extern "C" void mcq_wgrad_defer(int32_t on) {
    std::lock_guard<std::mutex> lock(g_defer_mu);
    g_defer = on != 0;
}

extern "C" int32_t mcq_wgrad_pending(void) {
    std::lock_guard<std::mutex> lock(g_defer_mu);
    const auto it = g_jobs.find(current_device());
    return it == g_jobs.end() ? 0 : (int32_t)it->second.size();
}

extern "C" int mcq_wgrad_flush(int32_t discard, void* stream) {
    std::vector<ReduceJob> jobs;
    {
        std::lock_guard<std::mutex> lock(g_defer_mu);
        if (discard) { g_jobs.clear(); return MCQ_OK; }            // (every device's: the pass they belonged to is gone)
        const auto it = g_jobs.find(current_device());
        if (it != g_jobs.end()) { jobs.swap(it->second); g_jobs.erase(it); }
    }
    if (jobs.empty()) return MCQ_OK;
    // a launch's grid.x is its LARGEST job's block count: jobs go largest first, and a launch ends where the next job would leave
    // more than a quarter of its row of blocks empty (the first form mixed 128 -> 512 shuffle convolutions with 1x1 ones:
    // three quarters of 738 k workgroups were dispatched to leave at once, 233 us)
    auto blocks_of = [](const ReduceJob& q) { return (unsigned)(((size_t)q.taps * q.Cout * q.Cin + (q.dbias ? (size_t)q.Cout : 0) + 63) / 64); };
    std::stable_sort(jobs.begin(), jobs.end(), [&](const ReduceJob& a, const ReduceJob& b) { return blocks_of(a) > blocks_of(b); });
    size_t at = 0;
    while (at < jobs.size()) {
        ReduceBatch b;
        const unsigned most = blocks_of(jobs[at]);
        int n = 0;
        while (n < REDUCE_BATCH && at + n < jobs.size() && 4u * blocks_of(jobs[at + n]) >= 3u * most) ++n;
        for (int j = 0; j < REDUCE_BATCH; ++j) b.job[j] = jobs[at + (j < n ? j : 0)];
        hipLaunchKernelGGL(wgrad_rows_reduce_batch_kernel, dim3(most, (unsigned)n), dim3(256), 0, (hipStream_t)stream, b);
        at += n;
    }
    return mcq_check_launch();
}
