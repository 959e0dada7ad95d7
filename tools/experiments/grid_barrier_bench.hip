// Cost of a device-wide barrier between a few persistent workgroups (one per CU) on MI355X: the building block of a
// fused-over-time recurrent kernel.  hipcc --offload-arch=gfx950 -O3 grid_barrier_bench.hip -o gbb && ./gbb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256) void bar_kernel(unsigned* counter, float* buf, int iters, int payload_floats, long long* cycles, int* err) {
    const int G = gridDim.x, tid = threadIdx.x;
    long long t0 = 0;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (it == 8 && tid == 0) t0 = wall_clock64();
        // every workgroup publishes `payload_floats` values, then reads everybody's
        for (int i = tid; i < payload_floats; i += 256) buf[((it & 1) * G + blockIdx.x) * payload_floats + i] = (float)(it + blockIdx.x);
        __syncthreads();
        if (tid == 0) {
            __threadfence();
            atomicAdd(counter, 1u);
            const unsigned target = (unsigned)G * (unsigned)(it + 1);
            int spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 22)) { *err = 1; break; }
            }
            __threadfence();
        }
        __syncthreads();
        for (int g = 0; g < G; ++g)
            for (int i = tid; i < payload_floats; i += 256) acc += buf[((it & 1) * G + g) * payload_floats + i];
    }
    if (tid == 0) cycles[blockIdx.x] = wall_clock64() - t0;
    if (acc == 12345.f) buf[0] = acc;
}
int main() {
    unsigned* counter; float* buf; long long* cyc; int* err;
    hipMalloc(&counter, 4); hipMalloc(&buf, 2 * 256 * 4096 * 4); hipMalloc(&cyc, 256 * 8); hipMalloc(&err, 4);
    for (int G : {8, 16, 32, 64, 128}) for (int payload : {64, 1024}) {
        hipMemset(counter, 0, 4); hipMemset(err, 0, 4);
        const int iters = 1008;
        hipLaunchKernelGGL(bar_kernel, dim3(G), dim3(256), 0, 0, counter, buf, iters, payload, cyc, err);
        hipDeviceSynchronize();
        long long h[256]; int e; hipMemcpy(h, cyc, G * 8, hipMemcpyDeviceToHost); hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
        long long mx = 0; for (int i = 0; i < G; ++i) mx = h[i] > mx ? h[i] : mx;
        printf("G=%3d payload %5d B per WG: %.2f us per iteration (wall_clock64 at 100 MHz)%s\n", G, payload * 4, mx / 100.0 / (iters - 8), e ? "  [spin limit hit]" : "");
    }
    return 0;
}
