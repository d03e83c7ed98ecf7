#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// where does "global_load_lds_dwordx4 voff, s[base] offset:IMM" read from and write to?
__global__ void k(const unsigned* base, unsigned* out) {
    __shared__ unsigned sm[2048];  // 8 KiB
    for (int i = threadIdx.x; i < 2048; i += 64) sm[i] = 0xdeadbeefu;
    __syncthreads();
    unsigned voff = threadIdx.x * 16;
    unsigned lds = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)sm;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\ts_waitcnt vmcnt(0)" ::"v"(voff), "s"(base), "s"(lds) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) out[i] = sm[i];
}
int main() {
    std::vector<unsigned> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = i;
    unsigned *d, *o;
    hipMalloc(&d, 4096 * 4); hipMalloc(&o, 2048 * 4);
    hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    std::vector<unsigned> r(2048);
    hipMemcpy(r.data(), o, 2048 * 4, hipMemcpyDeviceToHost);
    int first = -1;
    for (int i = 0; i < 2048; ++i) if (r[i] != 0xdeadbeefu) { first = i; break; }
    printf("first written dword index %d (byte %d), value %u (source byte %u); next %u %u %u; dword+4: %u\n", first, first * 4, r[first], r[first] * 4, r[first+1], r[first+2], r[first+3], r[first+4]);
    int cnt = 0; for (int i = 0; i < 2048; ++i) cnt += r[i] != 0xdeadbeefu;
    printf("dwords written %d\n", cnt);
    return 0;
}
