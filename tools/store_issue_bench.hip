// tools/store_issue_bench.hip -- what a vector store costs the wave that issues it (round 6).   GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/store_issue_bench.hip -o /tmp/store_issue_bench && /tmp/store_issue_bench
// One workgroup per CU (64 KB of LDS), W waves per workgroup (1, 2 or 4: one wave per SIMD), every wave issues N back-to-back stores of
// 4 / 8 / 16 bytes per lane to its own region (1 KB apart per instruction, streamed) and times them with s_memtime; then the same with
// EXEC = 0 (no lane enabled), and N loads of 16 bytes per lane for comparison.  Prints shader cycles per instruction (s_memtime runs at
// the shader clock; s_memrealtime, 100 MHz, gives the clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int N = 256;

// FILL > 0: that many independent v_add_f32 between two memory instructions -- does a store hold the wave's ISSUE (cycles add up) or only its
// memory pipeline (the VALU work hides under it)?
template <int DW, bool MASKED, bool LOAD, int FILL = 0>
__global__ __launch_bounds__(256) void k(unsigned* buf, unsigned long long* out, int waves) {
    __shared__ char pad[60000];
    if (threadIdx.x == 999) pad[0] = 1;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (w >= waves) return;
    char* base = (char*)buf + ((size_t)(blockIdx.x * 4 + w) * N) * 1024 + lane * (DW * 4);
    unsigned v0 = lane, v1 = lane + 1, v2 = lane + 2, v3 = lane + 3;
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0), "=s"(t0));
    if (MASKED) asm volatile("s_mov_b64 exec, 0");
#pragma unroll 16
    for (int i = 0; i < N; ++i) {
        char* p = base + (size_t)i * 1024;
        if (LOAD) {
            asm volatile("global_load_dwordx4 v[100:103], %0, off" ::"v"(p) : "v100", "v101", "v102", "v103", "memory");
        } else if (DW == 4) {
            asm volatile("global_store_dwordx4 %0, v[104:107], off nt" ::"v"(p) : "memory");
        } else if (DW == 2) {
            asm volatile("global_store_dwordx2 %0, v[104:105], off nt" ::"v"(p) : "memory");
        } else {
            asm volatile("global_store_dword %0, v104, off nt" ::"v"(p) : "memory");
        }
        if (FILL > 0) {
#pragma unroll
            for (int f = 0; f < FILL; ++f) asm volatile("v_add_f32 v[%0], 1.0, v[%0]" ::"n"(110 + (f & 15)) : "memory");
        }
    }
    if (MASKED) asm volatile("s_mov_b64 exec, -1");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));                    // ISSUE time: the last instruction has been issued
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1));
    if (lane == 0) { out[(blockIdx.x * 4 + w) * 2] = t1 - t0; out[(blockIdx.x * 4 + w) * 2 + 1] = r1 - r0; }
    (void)v0; (void)v1; (void)v2; (void)v3;
}

template <int DW, bool MASKED, bool LOAD, int FILL = 0>
void run(const char* what, unsigned* buf, unsigned long long* out, int cus) {
    for (int waves : {1, 2, 4}) {
        if (FILL > 0 && waves == 2) continue;
        hipLaunchKernelGGL((k<DW, MASKED, LOAD, FILL>), dim3(cus), dim3(256), 0, 0, buf, out, waves);
        hipLaunchKernelGGL((k<DW, MASKED, LOAD, FILL>), dim3(cus), dim3(256), 0, 0, buf, out, waves);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(cus * 8);
        (void)hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
        double cyc = 0, rt = 0;
        int n = 0;
        for (int b = 0; b < cus; ++b)
            for (int w = 0; w < waves; ++w) { cyc += h[(b * 4 + w) * 2]; rt += h[(b * 4 + w) * 2 + 1]; ++n; }
        cyc /= n; rt /= n;
        printf("%-34s %d wave(s) per CU: %7.1f shader cycles of issue per instruction  (%5.1f bytes per cycle and wave); all %d done in %6.1f us\n", what, waves,
               cyc / N, LOAD || MASKED ? 0.0 : 64.0 * DW * 4 / (cyc / N), N, rt / 100.0);
    }
}

int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    unsigned* buf;
    unsigned long long* out;
    (void)hipMalloc(&buf, (size_t)cus * 4 * N * 1024 + 4096);
    (void)hipMalloc(&out, cus * 8 * 8);
    (void)hipMemset(buf, 0, (size_t)cus * 4 * N * 1024);
    printf("%s, %d CUs; %d instructions per wave, every CU busy\n", prop.name, cus, N);
    run<4, false, false>("global_store_dwordx4 (1 KB)", buf, out, cus);
    run<2, false, false>("global_store_dwordx2 (512 B)", buf, out, cus);
    run<1, false, false>("global_store_dword (256 B)", buf, out, cus);
    run<4, true, false>("global_store_dwordx4, EXEC = 0", buf, out, cus);
    run<4, false, true>("global_load_dwordx4 (1 KB)", buf, out, cus);
    run<4, false, false, 8>("store x4 + 8 v_add_f32 (32 cyc)", buf, out, cus);
    run<4, false, false, 16>("store x4 + 16 v_add_f32 (64 cyc)", buf, out, cus);
    run<4, false, false, 32>("store x4 + 32 v_add_f32 (128 cyc)", buf, out, cus);
    run<4, true, false, 16>("EXEC=0 store x4 + 16 v_add (64)", buf, out, cus);
    run<4, false, true, 16>("load x4 + 16 v_add_f32 (64 cyc)", buf, out, cus);
    run<4, false, true, 32>("load x4 + 32 v_add_f32 (128 cyc)", buf, out, cus);
    return 0;
}
