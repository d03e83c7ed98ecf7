// Where do the workgroups of a 512-block launch (512 threads, 72 KiB LDS: two per CU) land?  Prints per-CU counts keyed by
// (XCC_ID, HW_ID[15:8]) and whether blocks [0,256) / [256,512) each cover every CU once.
// build+run: hipcc --offload-arch=gfx950 -O2 cu_placement_probe.hip -o /tmp/cup && /tmp/cup
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(512) void k(unsigned* out, int spin) {
    extern __shared__ char smem[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long t0 = clock64();
    while (clock64() - t0 < spin) { smem[threadIdx.x] = (char)t0; }
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc; }
}
int main() {
    const int NB = 512;
    unsigned* d; hipMalloc(&d, NB * 8);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 73728);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k, dim3(NB), dim3(512), 73728, 0, d, 200000);
        hipDeviceSynchronize();
    }
    std::vector<unsigned> h(NB * 2);
    hipMemcpy(h.data(), d, NB * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;
    for (int b = 0; b < NB; ++b) cu[((h[b * 2 + 1] & 0xf) << 8) | ((h[b * 2] >> 8) & 0xff)].push_back(b);
    int ok_first = 0, two = 0;
    for (auto& kv : cu) {
        int first = 0;
        for (int b : kv.second) first += b < 256;
        ok_first += first == 1; two += kv.second.size() == 2;
    }
    printf("distinct CUs %zu, CUs holding exactly two blocks %d, CUs holding exactly one block of [0,256) %d\n", cu.size(), two, ok_first);
    int shown = 0;
    for (auto& kv : cu) { if (shown++ >= 12) break; printf("  key %03x:", kv.first); for (int b : kv.second) printf(" %d", b); printf("\n"); }
    printf("block 0..15 hw_id:"); for (int b = 0; b < 16; ++b) printf(" %08x/%x", h[b * 2], h[b * 2 + 1] & 0xf); printf("\n");
    return 0;
}
