// Does the gfx950 raw-buffer range check include the SGPR offset?  Loads element (voffset + soffset) of a 64-float
// buffer whose descriptor covers only the first 16 floats; prints what comes back for in-range / out-of-range sums.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* x, float* out, int soff) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, 16 * 4, 0x00020000);
    const int lane = threadIdx.x;
    // voffset always in range (lane & 3 -> 0..12 bytes), soffset pushes some accesses past num_records
    out[lane] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (lane & 3) * 4, soff, 0));
    out[64 + lane] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (lane & 3) * 4 + soff, 0, 0));
}
int main() {
    float h[64], *d, *o, ho[128];
    for (int i = 0; i < 64; ++i) h[i] = 100.0f + i;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (int soff : {0, 32, 60, 64, 128}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, soff);
        hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
        printf("soffset=%3d bytes: via soffset -> %.0f %.0f %.0f %.0f | via voffset -> %.0f %.0f %.0f %.0f\n", soff, ho[0], ho[1], ho[2], ho[3],
               ho[64], ho[65], ho[66], ho[67]);
    }
    return 0;
}
