// tools/cumask_probe.hip -- which physical CUs does bit i of a hipExtStreamCreateWithCUMask mask name?  (GPU box)
//   hipcc --offload-arch=gfx950 -O2 tools/cumask_probe.hip -o /tmp/cumask_probe && /tmp/cumask_probe [first cus]...
// For each (first, cus) pair: a stream restricted to logical CUs first .. first+cus-1, a grid of 4096 single-wave workgroups that
// spin for a few microseconds, and the set of (XCC_ID, SE_ID, SH_ID, CU_ID) they ran on.  ops.MaskedStream / DESIGN.md 2.8 rely on
// "consecutive bits go round the XCDs first": a run of 8k bits must show k CUs on each of the 8 XCDs, and two disjoint runs
// disjoint CU sets.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>

__global__ void where_kernel(unsigned* out) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) {}                  // 20 us at 100 MHz: the grid has to spread over every CU it may use
    if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 0xf) << 16) | ((hw >> 8) & 0xffu);      // bits 7:0 = CU_ID[3:0] | SH_ID << 4 | SE_ID << 5
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int total = prop.multiProcessorCount, words = (total + 31) / 32;
    std::vector<std::pair<int, int>> runs;
    for (int i = 1; i + 1 < argc; i += 2) runs.push_back({atoi(argv[i]), atoi(argv[i + 1])});
    // (a mask that leaves an XCD without any CU does not restrict that XCD at all -- bits 0..0 run on 1 + 7 x 32 CUs: masks must
    //  name at least one CU of every XCD, which multiples of 8 consecutive bits do)
    if (runs.empty()) runs = {{0, 8}, {0, 64}, {64, 192}, {248, 8}, {0, 1}, {8, 16}};
    const int N = 4096;
    unsigned* d;
    (void)hipMalloc(&d, N * 4);
    std::vector<unsigned> h(N);
    std::vector<std::set<unsigned>> seen;
    printf("%s: %d CUs, mask words %d\n", prop.name, total, words);
    for (auto [first, cus] : runs) {
        std::vector<uint32_t> mask(words, 0);
        for (int i = first; i < first + cus && i < total; ++i) mask[i / 32] |= 1u << (i % 32);
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, words, mask.data()) != hipSuccess) { printf("create failed\n"); return 1; }
        hipLaunchKernelGGL(where_kernel, dim3(N), dim3(64), 0, s, d);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h.data(), d, N * 4, hipMemcpyDeviceToHost);
        std::set<unsigned> cu(h.begin(), h.end());
        std::map<unsigned, int> per_xcc;
        for (unsigned c : cu) per_xcc[c >> 16]++;
        printf("bits %3d..%3d (%3d CUs): ran on %3zu distinct CUs; per XCC:", first, first + cus - 1, cus, cu.size());
        for (auto [x, n] : per_xcc) printf(" %u:%d", x, n);
        if (cu.size() <= 4) for (unsigned c : cu) printf("  [xcc %u se %u sh %u cu %u]", c >> 16, (c >> 5) & 7, (c >> 4) & 1, c & 15);
        for (size_t j = 0; j < seen.size(); ++j) {
            size_t common = 0;
            for (unsigned c : cu) common += seen[j].count(c);
            if (common) printf("  | %zu in common with run %zu", common, j);
        }
        printf("\n");
        seen.push_back(cu);
        (void)hipStreamDestroy(s);
    }
    return 0;
}
