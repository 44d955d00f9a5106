// Validation metrics either side of the Compressor path (SURVEY.md §8 "next" row 3), gfx950.
//
//   mcq_ms_ssim_u8      MS-SSIM of two uint8 image batches exactly as the reference's validator computes it
//                       (mcquic/validate/handlers.py:14-27 -> mcquic/validate/metrics.py:69-104, 142-193):
//                       images .float(), data range 255, 11-tap sigma-1.5 Gaussian 'valid' blur (rows first, then
//                       columns) of x, y, x^2, y^2, xy, per-(image, channel) means of the SSIM and contrast-structure
//                       maps on five levels, 2x2 average pooling with zero padding of odd sides counted in the divisor,
//                       prod_l relu(v_l)^w_l, mean over channels.
//   mcq_sqdiff_sum_u8   exact integer sum of squared differences per image (the float64 MSE of metrics.py:271-274 is
//                       this sum / count, bit for bit).
//
// One workgroup blurs a 16 x 118 output tile: the 26 x 128 input patch of x and y goes to LDS once, the vertical pass
// keeps an 18-row column strip in registers (8 outputs x 5 moments per thread), the horizontal pass reads the five
// moment planes back with lanes along the row (conflict-free).  Every fp32 operation follows the order of
// oracle/metrics_ref.py (taps added in index order, multiply and add rounded separately: the library is built with
// -ffp-contract=off); the map means are accumulated in float64 and reduced in a fixed order, so results are
// deterministic and differ from the CPU restatement only in the rounding of those means.
#include "mcq_common.h"
#include "../../include/mcquic_hip.h"

namespace {

constexpr int TH = 16, TW = 118, IH = TH + 10, IW = 128;

// float32 window of metrics.py:22-37 for size 11, sigma 1.5 (values as torch computes them; checked against the
// oracle's gauss_window() in tests/test_host_abi.py through mcq_ms_ssim_window)
#define MCQ_WIN_VALUES                                                                                              \
    { 0x1.0d957p-10f, 0x1.f1fe02p-8f, 0x1.26eb18p-5f, 0x1.bff0fep-4f, 0x1.b43c3ep-3f, 0x1.10656p-2f, 0x1.b43c3ep-3f, \
      0x1.bff0fep-4f, 0x1.26eb18p-5f, 0x1.f1fe02p-8f, 0x1.0d957p-10f }
__constant__ float c_win[11] = MCQ_WIN_VALUES;
const float h_win[11] = MCQ_WIN_VALUES;
// metrics.py:19 level weights, float32
__constant__ float c_level_w[5] = {0x1.6f0068p-5f, 0x1.247454p-2f, 0x1.334d6ap-2f, 0x1.e3f142p-3f, 0x1.10ff98p-3f};
constexpr float C1 = 6.5025f;      // (0.01 * 255)^2 rounded to float32 where it meets the float32 maps
constexpr float C2 = 58.5225f;     // (0.03 * 255)^2

__device__ __forceinline__ float to_f(uint8_t v) { return (float)v; }
__device__ __forceinline__ float to_f(float v) { return v; }

template <typename T>
__global__ __launch_bounds__(256) void ssim_level_kernel(const T* __restrict__ X, const T* __restrict__ Y, int H, int W,
                                                         int Ho, int Wo, double* __restrict__ partial) {
    // 41 KB: the input patches (2 x 26 x 128) and, once they are consumed, the five moment planes over the same bytes
    __shared__ float smem[5 * TH * (IW + 1)];
    __shared__ double red[2][4];
    float (*sx)[IW] = reinterpret_cast<float (*)[IW]>(smem);
    float (*sy)[IW] = reinterpret_cast<float (*)[IW]>(smem + IH * IW);
    float (*sv)[TH][IW + 1] = reinterpret_cast<float (*)[TH][IW + 1]>(smem);
    static_assert(2 * IH * IW <= 5 * TH * (IW + 1), "patches fit under the moment planes");
    const int tid = threadIdx.x;
    const size_t plane = blockIdx.z;
    const int r0 = blockIdx.y * TH, c0 = blockIdx.x * TW;
    const T* xp = X + plane * (size_t)H * W;
    const T* yp = Y + plane * (size_t)H * W;

    for (int i = tid; i < IH * IW; i += 256) {
        const int r = i >> 7, c = i & (IW - 1);
        const int gr = r0 + r, gc = c0 + c;
        const bool ok = gr < H && gc < W;
        const size_t o = (size_t)gr * W + gc;
        sx[r][c] = ok ? to_f(xp[o]) : 0.0f;
        sy[r][c] = ok ? to_f(yp[o]) : 0.0f;
    }
    __syncthreads();

    {   // vertical pass: column c, output rows rg*8 .. rg*8+7 from input rows rg*8 .. rg*8+17
        const int c = tid & (IW - 1), rg = tid >> 7;
        float a[5][8];
#pragma unroll
        for (int t = 0; t < 18; ++t) {
            const float x = sx[rg * 8 + t][c], y = sy[rg * 8 + t][c];
            const float q[5] = {x, y, x * x, y * y, x * y};
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                const int tap = t - o;
                if (tap >= 0 && tap < 11) {
#pragma unroll
                    for (int k = 0; k < 5; ++k) a[k][o] = tap == 0 ? c_win[0] * q[k] : a[k][o] + c_win[tap] * q[k];
                }
            }
        }
        __syncthreads();                     // every strip is in registers: the patches may be overwritten
#pragma unroll
        for (int k = 0; k < 5; ++k)
#pragma unroll
            for (int o = 0; o < 8; ++o) sv[k][rg * 8 + o][c] = a[k][o];
    }
    __syncthreads();

    double s_ssim = 0.0, s_cs = 0.0;
    for (int idx = tid; idx < TH * TW; idx += 256) {
        const int r = idx / TW, c = idx - r * TW;
        float f[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            float acc = c_win[0] * sv[k][r][c];
#pragma unroll
            for (int t = 1; t < 11; ++t) acc = acc + c_win[t] * sv[k][r][c + t];
            f[k] = acc;
        }
        const float mu1_sq = f[0] * f[0], mu2_sq = f[1] * f[1], mu12 = f[0] * f[1];
        const float s1 = f[2] - mu1_sq, s2 = f[3] - mu2_sq, s12 = f[4] - mu12;
        const float cs = (2.0f * s12 + C2) / (s1 + s2 + C2);
        const float ss = ((2.0f * mu12 + C1) / (mu1_sq + mu2_sq + C1)) * cs;
        if (r0 + r < Ho && c0 + c < Wo) {
            s_ssim += (double)ss;
            s_cs += (double)cs;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s_ssim += __shfl_down(s_ssim, off);
        s_cs += __shfl_down(s_cs, off);
    }
    if ((tid & 63) == 0) {
        red[0][tid >> 6] = s_ssim;
        red[1][tid >> 6] = s_cs;
    }
    __syncthreads();
    if (tid == 0) {
        const size_t tile = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        const size_t ntile = (size_t)gridDim.x * gridDim.y;
        double* o = partial + (plane * ntile + tile) * 2;
        o[0] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
        o[1] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
    }
}

// per plane: tile partials in tile order -> means of the two maps: level_out[plane] = {ssim, cs}
__global__ void ssim_finish_kernel(const double* __restrict__ partial, int planes, int ntile, double count,
                                   float* __restrict__ level_out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= planes) return;
    double a = 0.0, b = 0.0;
    const double* q = partial + (size_t)p * ntile * 2;
    for (int t = 0; t < ntile; ++t) {
        a += q[2 * t];
        b += q[2 * t + 1];
    }
    level_out[2 * p] = (float)(a / count);
    level_out[2 * p + 1] = (float)(b / count);
}

// metrics.py:177-179: avg_pool2d(kernel 2, stride 2, padding = side % 2), padded zeros counted in the divisor
template <typename T>
__global__ void halve_kernel(const T* __restrict__ X, const T* __restrict__ Y, float* __restrict__ Xo,
                             float* __restrict__ Yo, int H, int W, int Ho, int Wo, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int xo = (int)(i % Wo);
    const size_t t = i / Wo;
    const int yo = (int)(t % Ho);
    const size_t plane = t / Ho;
    const int y0 = 2 * yo - (H & 1), x0 = 2 * xo - (W & 1);
    const T* xp = X + plane * (size_t)H * W;
    const T* yp = Y + plane * (size_t)H * W;
    float vx[4], vy[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int yy = y0 + (k >> 1), xx = x0 + (k & 1);
        const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
        const size_t o = (size_t)yy * W + xx;
        vx[k] = ok ? to_f(xp[o]) : 0.0f;
        vy[k] = ok ? to_f(yp[o]) : 0.0f;
    }
    Xo[i] = (((vx[0] + vx[1]) + vx[2]) + vx[3]) / 4.0f;
    Yo[i] = (((vy[0] + vy[1]) + vy[2]) + vy[3]) / 4.0f;
}

// metrics.py:184-191: prod over levels of relu(cs_l)^w_l (relu(ssim)^w on the last level), mean over channels
__global__ void ms_ssim_combine_kernel(const float* __restrict__ levels, int N, int C, float* __restrict__ out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int planes = N * C;
    float sum = 0.0f;
    for (int c = 0; c < C; ++c) {
        float prod = 1.0f;
        for (int l = 0; l < 5; ++l) {
            const float* lv = levels + ((size_t)l * planes + (size_t)n * C + c) * 2;
            const float v = fmaxf(l < 4 ? lv[1] : lv[0], 0.0f);
            const float pw = powf(v, c_level_w[l]);
            prod = l == 0 ? pw : prod * pw;
        }
        sum = c == 0 ? prod : sum + prod;
    }
    out[n] = sum / (float)C;
}

// (zeroing is a kernel, not hipMemsetAsync: a memset NODE of a captured hipGraph is replayed wrongly by ROCm 7.2 after eager blit
//  work, see csrc/train_ops.hip)
__global__ void zero_i64_kernel(int64_t* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = 0;
}

constexpr int SQ_CHUNK = 16384;       // bytes per workgroup (64 per thread: the uint32 partial cannot overflow)
__global__ __launch_bounds__(256) void sqdiff_sum_u8_kernel(const uint8_t* __restrict__ x, const uint8_t* __restrict__ y,
                                                            int64_t per_image, int vec4, unsigned long long* __restrict__ out) {
    const int n = blockIdx.y;
    const int64_t base = (int64_t)blockIdx.x * SQ_CHUNK;
    const uint8_t* xp = x + (size_t)n * per_image;
    const uint8_t* yp = y + (size_t)n * per_image;
    const int64_t end = base + SQ_CHUNK < per_image ? base + SQ_CHUNK : per_image;
    unsigned acc = 0;
    if (vec4) {                           // per_image % 16 == 0 and 16-byte aligned bases: 16 pixels per load
        for (int64_t i = base + threadIdx.x * 16; i < end; i += 256 * 16) {
            const uint4 a = *reinterpret_cast<const uint4*>(xp + i);
            const uint4 b = *reinterpret_cast<const uint4*>(yp + i);
            const unsigned aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int sft = 0; sft < 32; sft += 8) {
                    const int d = (int)((aw[k] >> sft) & 255u) - (int)((bw[k] >> sft) & 255u);
                    acc += (unsigned)(d * d);
                }
        }
    } else {
        for (int64_t i = base + threadIdx.x; i < end; i += 256) {
            const int d = (int)xp[i] - (int)yp[i];
            acc += (unsigned)(d * d);
        }
    }
    unsigned long long s = acc;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    __shared__ unsigned long long red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out + n, red[0] + red[1] + red[2] + red[3]);   // integer: order-independent
}

struct Pyramid {
    int H[5], W[5];
    size_t plane_floats[5];     // level l image plane size (level 0 is the uint8 input, not stored)
    int tiles_x[5], tiles_y[5];
};

inline bool make_pyramid(int H, int W, Pyramid& p) {
    if (H <= 160 || W <= 160) return false;        // metrics.py:163-166
    for (int l = 0; l < 5; ++l) {
        p.H[l] = H;
        p.W[l] = W;
        p.plane_floats[l] = (size_t)H * W;
        const int Ho = H - 10, Wo = W - 10;
        p.tiles_x[l] = (Wo + TW - 1) / TW;
        p.tiles_y[l] = (Ho + TH - 1) / TH;
        const int ph = H & 1, pw = W & 1;
        H = (H + 2 * ph - 2) / 2 + 1;
        W = (W + 2 * pw - 2) / 2 + 1;
    }
    return true;
}

struct Workspace {
    size_t x_off[5], y_off[5];   // float offsets of the pooled images (levels 1..4)
    size_t partial_off;          // byte offset of the double partials (8-aligned)
    size_t levels_off;           // byte offset of float level_out[5][planes][2]
    size_t bytes;
};

inline Workspace layout(const Pyramid& p, size_t planes) {
    Workspace w{};
    size_t fl = 0;
    for (int l = 1; l < 5; ++l) {
        w.x_off[l] = fl;
        fl += planes * p.plane_floats[l];
        w.y_off[l] = fl;
        fl += planes * p.plane_floats[l];
    }
    size_t b = (fl * sizeof(float) + 7) & ~(size_t)7;
    w.partial_off = b;
    size_t max_tiles = 0;
    for (int l = 0; l < 5; ++l) max_tiles = max_tiles > (size_t)p.tiles_x[l] * p.tiles_y[l] ? max_tiles : (size_t)p.tiles_x[l] * p.tiles_y[l];
    b += planes * max_tiles * 2 * sizeof(double);
    w.levels_off = b;
    b += 5 * planes * 2 * sizeof(float);
    w.bytes = b;
    return w;
}

}  // namespace

extern "C" void mcq_ms_ssim_window(float* out11) {
    for (int i = 0; i < 11; ++i) out11[i] = h_win[i];
}

extern "C" size_t mcq_ms_ssim_workspace_bytes(int32_t N, int32_t C, int32_t H, int32_t W) {
    Pyramid p;
    if (N <= 0 || C <= 0 || !make_pyramid(H, W, p)) return 0;
    return layout(p, (size_t)N * C).bytes;
}

extern "C" int mcq_ms_ssim_u8(const uint8_t* x, const uint8_t* y, float* out, void* workspace, int32_t N, int32_t C,
                              int32_t H, int32_t W, void* stream) {
    if (!x || !y || !out || !workspace || N <= 0 || C <= 0) return MCQ_EINVAL;
    Pyramid p;
    if (!make_pyramid(H, W, p)) return MCQ_EINVAL;
    const size_t planes = (size_t)N * C;
    if (planes > 65535) return MCQ_ETOOLARGE;      // grid.z
    const Workspace ws = layout(p, planes);
    hipStream_t s = (hipStream_t)stream;
    char* base = (char*)workspace;
    float* fbase = (float*)workspace;
    double* partial = (double*)(base + ws.partial_off);
    float* levels = (float*)(base + ws.levels_off);
    for (int l = 0; l < 5; ++l) {
        const int Hl = p.H[l], Wl = p.W[l], Ho = Hl - 10, Wo = Wl - 10;
        const dim3 grid((unsigned)p.tiles_x[l], (unsigned)p.tiles_y[l], (unsigned)planes);
        const int ntile = p.tiles_x[l] * p.tiles_y[l];
        if (l == 0) hipLaunchKernelGGL(ssim_level_kernel<uint8_t>, grid, dim3(256), 0, s, x, y, Hl, Wl, Ho, Wo, partial);
        else hipLaunchKernelGGL(ssim_level_kernel<float>, grid, dim3(256), 0, s, (const float*)(fbase + ws.x_off[l]),
                                (const float*)(fbase + ws.y_off[l]), Hl, Wl, Ho, Wo, partial);
        hipLaunchKernelGGL(ssim_finish_kernel, dim3((unsigned)((planes + 63) / 64)), dim3(64), 0, s, (const double*)partial,
                           (int)planes, ntile, (double)Ho * (double)Wo, levels + (size_t)l * planes * 2);
        if (l < 4) {
            const size_t total = planes * p.plane_floats[l + 1];
            const dim3 g((unsigned)((total + 255) / 256));
            if (l == 0) hipLaunchKernelGGL(halve_kernel<uint8_t>, g, dim3(256), 0, s, x, y, fbase + ws.x_off[1], fbase + ws.y_off[1],
                                           Hl, Wl, p.H[1], p.W[1], total);
            else hipLaunchKernelGGL(halve_kernel<float>, g, dim3(256), 0, s, (const float*)(fbase + ws.x_off[l]),
                                    (const float*)(fbase + ws.y_off[l]), fbase + ws.x_off[l + 1], fbase + ws.y_off[l + 1], Hl, Wl,
                                    p.H[l + 1], p.W[l + 1], total);
        }
    }
    hipLaunchKernelGGL(ms_ssim_combine_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, s, (const float*)levels, N, C, out);
    return mcq_check_launch();
}

extern "C" int mcq_sqdiff_sum_u8(const uint8_t* x, const uint8_t* y, int64_t* out, int64_t per_image, int32_t N,
                                 void* stream) {
    if (!x || !y || !out || per_image <= 0 || N <= 0 || N > 65535) return MCQ_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(zero_i64_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, out, (int)N);
    const dim3 grid((unsigned)((per_image + SQ_CHUNK - 1) / SQ_CHUNK), (unsigned)N);
    const int vec4 = per_image % 16 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0;
    hipLaunchKernelGGL(sqdiff_sum_u8_kernel, grid, dim3(256), 0, s, x, y, per_image, vec4, (unsigned long long*)out);
    return mcq_check_launch();
}

// ---- self-test of the launch-error path ----------------------------------------------------------------------------
namespace { __global__ void mcq_noop_kernel() {} }

extern "C" int mcq_selftest_launch_failure(void* stream) {
    // 4096 threads per workgroup is beyond the device limit (1024): the launch is refused, hipGetLastError() reports it,
    // and mcq_check_launch() turns that into MCQ_ELAUNCH like for any real kernel (the error is not sticky)
    hipLaunchKernelGGL(mcq_noop_kernel, dim3(1), dim3(4096), 0, (hipStream_t)stream);
    return mcq_check_launch();
}

