// What a grid-wide barrier between two dependent layers would cost inside ONE persistent launch (instead of a kernel boundary):
// every workgroup writes a block of floats, releases it at agent scope, meets the others on an atomic counter, acquires, and
// reads the block its neighbour (on another XCD) wrote.  Prints the time per round and checks the data.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/grid_barrier.hip -o /tmp/grid_barrier && timeout 60 /tmp/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void rounds_kernel(float* buf, unsigned* counter, int rounds, int floats_per_wg, int* bad) {
    const unsigned nwg = gridDim.x;
    const unsigned wg = blockIdx.x;
    const unsigned peer = (wg + 1) % nwg;                         // linear neighbour = another XCD (ids are dealt round-robin)
    for (int r = 0; r < rounds; ++r) {
        float* mine = buf + ((size_t)(r & 1) * nwg + wg) * floats_per_wg;
        for (int i = threadIdx.x; i < floats_per_wg; i += blockDim.x) mine[i] = (float)(r * 131 + (int)wg + i);
        __threadfence();                                           // release at agent scope (L2 write-back across XCDs)
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(r + 1) * nwg;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        __threadfence();                                           // acquire side for the other lanes
        const float* theirs = buf + ((size_t)(r & 1) * nwg + peer) * floats_per_wg;
        for (int i = threadIdx.x; i < floats_per_wg; i += blockDim.x) {
            const float v = __builtin_nontemporal_load(theirs + i);
            if (v != (float)(r * 131 + (int)peer + i)) atomicAdd(bad, 1);
        }
    }
}

// The same round with a barrier built for the machine: arrivals counted per XCD (workgroup id & 7: the dispatcher deals workgroups
// round-robin to the 8 XCDs), the last arriver of an XCD bumps ONE global epoch counter, everybody polls the epoch with a relaxed
// agent-scope load and a longer sleep (256 pollers on one line are what made the naive form slow), one release fence before the
// arrival and one acquire fence after the wait.  `payload` = 0 measures the barrier alone.
__global__ __launch_bounds__(256) void rounds2_kernel(float* buf, unsigned* xcd_count, unsigned* epoch, int rounds, int floats_per_wg, int* bad, int payload) {
    const unsigned nwg = gridDim.x, wg = blockIdx.x, xcd = wg & 7u;
    const unsigned per_xcd = (nwg + 7u - xcd) / 8u;                // workgroups with this id & 7
    const unsigned peer = (wg + 1) % nwg;
    for (int r = 0; r < rounds; ++r) {
        float* mine = buf + ((size_t)(r & 1) * nwg + wg) * floats_per_wg;
        if (payload)
            for (int i = threadIdx.x; i < floats_per_wg; i += blockDim.x) mine[i] = (float)(r * 131 + (int)wg + i);
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            const unsigned got = __hip_atomic_fetch_add(xcd_count + xcd * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (got + 1 == (unsigned)(r + 1) * per_xcd) __hip_atomic_fetch_add(epoch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(r + 1) * 8u;
            while (__hip_atomic_load(epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(8);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (payload) {
            const float* theirs = buf + ((size_t)(r & 1) * nwg + peer) * floats_per_wg;
            for (int i = threadIdx.x; i < floats_per_wg; i += blockDim.x) {
                const float v = __builtin_nontemporal_load(theirs + i);
                if (v != (float)(r * 131 + (int)peer + i)) atomicAdd(bad, 1);
            }
        }
    }
}

__global__ void empty_kernel(float* buf) { if (threadIdx.x == 12345) buf[0] = 1.0f; }

int main() {
    const int nwg = 256, rounds = 200;
    for (int floats_per_wg : {256, 4096, 32768}) {
        float* buf; unsigned* counter; int* bad;
        hipMalloc(&buf, (size_t)2 * nwg * floats_per_wg * sizeof(float));
        hipMalloc(&counter, sizeof(unsigned)); hipMalloc(&bad, sizeof(int));
        hipMemset(counter, 0, sizeof(unsigned)); hipMemset(bad, 0, sizeof(int));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(rounds_kernel, dim3(nwg), dim3(256), 0, 0, buf, counter, 2, floats_per_wg, bad);   // warm-up (2 rounds)
        hipDeviceSynchronize();
        hipMemset(counter, 0, sizeof(unsigned));
        hipEventRecord(e0);
        hipLaunchKernelGGL(rounds_kernel, dim3(nwg), dim3(256), 0, 0, buf, counter, rounds, floats_per_wg, bad);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        int hbad = 0; hipMemcpy(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost);
        // the same amount of work as dependent launches: one kernel per round
        hipEventRecord(e0);
        for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(empty_kernel, dim3(nwg), dim3(256), 0, 0, buf);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms2 = 0; hipEventElapsedTime(&ms2, e0, e1);
        printf("%6d floats per workgroup (%5.1f MB per round): %.2f us per in-kernel round (write + barrier + read), stale reads %d; "
               "%.2f us per empty dependent launch\n", floats_per_wg, nwg * floats_per_wg * 4.0 / 1e6, ms * 1e3 / rounds, hbad, ms2 * 1e3 / rounds);
        for (int payload = 1; payload >= 0; --payload) {
            unsigned* xc; hipMalloc(&xc, 8 * 32 * sizeof(unsigned));
            hipMemset(xc, 0, 8 * 32 * sizeof(unsigned)); hipMemset(counter, 0, sizeof(unsigned)); hipMemset(bad, 0, sizeof(int));
            hipLaunchKernelGGL(rounds2_kernel, dim3(nwg), dim3(256), 0, 0, buf, xc, counter, 2, floats_per_wg, bad, payload);
            hipDeviceSynchronize();
            hipMemset(xc, 0, 8 * 32 * sizeof(unsigned)); hipMemset(counter, 0, sizeof(unsigned)); hipMemset(bad, 0, sizeof(int));
            hipEventRecord(e0);
            hipLaunchKernelGGL(rounds2_kernel, dim3(nwg), dim3(256), 0, 0, buf, xc, counter, rounds, floats_per_wg, bad, payload);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost);
            printf("       hierarchical barrier, %s: %.2f us per round, stale reads %d\n", payload ? "write + barrier + read" : "barrier alone", ms * 1e3 / rounds, hbad);
            hipFree(xc);
        }
        hipFree(buf); hipFree(counter); hipFree(bad);
    }
    return 0;
}
