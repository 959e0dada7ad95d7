// Practical MFMA ceiling of the chip under load (diagnostic, not part of the product):
//   a loop of nothing but v_mfma_f32_32x32x16_bf16 on register operands (random bf16 data), one / two waves per SIMD on
//   every CU, for long enough that the clock settles.  Prints TFLOP/s, the in-kernel clock (s_memtime / s_memrealtime) and
//   the same with all-zero operands (which draw less power and hold a higher clock).
// build + run:  hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_ceiling tools/mfma_ceiling.hip && /tmp/mfma_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int ACCS>
__global__ __launch_bounds__(256) void mfma_loop(const u32x4* __restrict__ src, float* __restrict__ out, int iters,
                                                 unsigned long long* __restrict__ clk) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    bf16x8 a[4], b[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(bf16x8, src[(tid * 6 + i) & 4095]);
#pragma unroll
    for (int i = 0; i < 2; ++i) b[i] = __builtin_bit_cast(bf16x8, src[(tid * 6 + 4 + i) & 4095]);
    f32x16 acc[ACCS];
#pragma unroll
    for (int i = 0; i < ACCS; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ACCS; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[i & 1], acc[i], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ACCS; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[tid] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

typedef __attribute__((ext_vector_type(4))) float f32x4v;
template <int ACCS>
__global__ __launch_bounds__(256) void mfma_loop16(const u32x4* __restrict__ src, float* __restrict__ out, int iters,
                                                   unsigned long long* __restrict__ clk) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = __builtin_bit_cast(bf16x8, src[(tid * 8 + i) & 4095]); b[i] = __builtin_bit_cast(bf16x8, src[(tid * 8 + 4 + i) & 4095]); }
    f32x4v acc[ACCS];
#pragma unroll
    for (int i = 0; i < ACCS; ++i) acc[i] = f32x4v{0.f, 0.f, 0.f, 0.f};
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ACCS; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ACCS; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[tid] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    std::vector<uint32_t> h(4096 * 4);
    std::mt19937 rng(3);
    u32x4* src; float* out; unsigned long long* clk;
    hipMalloc(&src, h.size() * 4); hipMalloc(&out, (size_t)cus * 2 * 256 * 4); hipMalloc(&clk, (size_t)cus * 2 * 16);
    for (int zero = 0; zero < 2; ++zero) {
        for (auto& x : h) {
            // two bf16 values in [-1, 1)
            auto bf = [&]() { float f = (rng() & 0xffff) / 32768.f - 1.f; uint32_t u; __builtin_memcpy(&u, &f, 4); return u >> 16; };
            x = zero ? 0u : (bf() | (bf() << 16));
        }
        hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (int wps = 1; wps <= 2; ++wps) {                 // waves per SIMD: one or two 256-thread workgroups per CU
            const int grid = cus * wps, iters = 200000;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL((mfma_loop<8>), dim3(grid), dim3(256), 0, 0, src, out, 2000, clk);      // warm-up
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL((mfma_loop<8>), dim3(grid), dim3(256), 0, 0, src, out, iters, clk);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> c((size_t)grid * 2);
            hipMemcpy(c.data(), clk, c.size() * 8, hipMemcpyDeviceToHost);
            double ghz = 0;
            for (int i = 0; i < grid; ++i) ghz += (double)c[2 * i] / ((double)c[2 * i + 1] * 10.0);       // memrealtime = 100 MHz
            ghz /= grid;
            const double flops = (double)grid * 4 * iters * 8 * 32768.0;
            printf("32x32x16 %-7s operands, %d wave(s)/SIMD on %d CUs: %7.1f TFLOP/s  (%.1f ms, in-kernel clock %.2f GHz, %.1f %% of 2.5 PF)\n",
                   zero ? "zero" : "random", wps, cus, flops / ms / 1e9, ms, ghz, flops / ms / 1e9 / 25.0);
            // the 16x16x32 shape: 16 accumulators of 4 registers, the same flops per instruction pair
            hipLaunchKernelGGL((mfma_loop16<16>), dim3(grid), dim3(256), 0, 0, src, out, 2000, clk);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL((mfma_loop16<16>), dim3(grid), dim3(256), 0, 0, src, out, iters, clk);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(c.data(), clk, c.size() * 8, hipMemcpyDeviceToHost);
            ghz = 0;
            for (int i = 0; i < grid; ++i) ghz += (double)c[2 * i] / ((double)c[2 * i + 1] * 10.0);
            ghz /= grid;
            const double flops16 = (double)grid * 4 * iters * 16 * 16384.0;
            printf("16x16x32 %-7s operands, %d wave(s)/SIMD on %d CUs: %7.1f TFLOP/s  (%.1f ms, in-kernel clock %.2f GHz, %.1f %% of 2.5 PF)\n",
                   zero ? "zero" : "random", wps, cus, flops16 / ms / 1e9, ms, ghz, flops16 / ms / 1e9 / 25.0);
        }
    }
    return 0;
}
