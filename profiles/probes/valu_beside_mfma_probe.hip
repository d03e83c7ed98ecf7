// Round 5 probe: what does one filler instruction cost a wave that issues v_mfma_f32_32x32x16_f16 back to back (one wave per SIMD)?
// Each wave runs ITERS x { 8 MFMAs (two accumulators), N fillers of kind T on independent registers behind each }; cycles per iteration from s_memtime.
// build + run (GPU box): hipcc -O3 -mllvm -amdgpu-mfma-vgpr-form=1 --offload-arch=gfx950 valu_beside_mfma_probe.hip -o /tmp/vprobe && /tmp/vprobe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int T>
__device__ __forceinline__ void filler(uint32_t& d, uint32_t a, uint32_t b, float fs) {
    if constexpr (T == 0) asm volatile("v_and_b32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    else if constexpr (T == 1) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(a));
    else if constexpr (T == 2) asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    else if constexpr (T == 3) asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    else if constexpr (T == 4) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "+v"(d) : "v"(a), "v"(fs), "v"(fs));
    else if constexpr (T == 5) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(a));
    else if constexpr (T == 6) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(a));
    else if constexpr (T == 7) asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(a));
    else if constexpr (T == 8) asm volatile("v_lshrrev_b32 %0, 4, %1" : "=v"(d) : "v"(a));
    else if constexpr (T == 9) asm volatile("v_mul_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    else if constexpr (T == 10) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(d) : "v"(a));
    else if constexpr (T == 11) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    else if constexpr (T == 12) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    else if constexpr (T == 13) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(d) : "v"(a));
    else if constexpr (T == 14) asm volatile("v_bfe_u32 %0, %1, 4, 4" : "=v"(d) : "v"(a));
}

template <int T, int N, int MF>
__global__ __launch_bounds__(256) void probe(uint32_t* out, uint64_t* cyc, int iters) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f - i * 0.01f); }
    f16v acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    uint32_t r[16];
    for (int i = 0; i < 16; ++i) r[i] = threadIdx.x * 7 + i;
    uint32_t s0 = 0x3c003c00u + threadIdx.x, s1 = 0x0f0f0f0fu;
    float fs = 1.5f;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {                 // 8 MFMAs per iteration, N fillers behind each
            if (MF) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < N; ++i) filler<T>(r[i & 15], s0, s1, fs);
            if (MF) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < N; ++i) filler<T>(r[(i + 8) & 15], s0, s1, fs);
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    uint32_t x = 0;
    for (int i = 0; i < 16; ++i) x ^= r[i];
    float sacc = 0;
    for (int i = 0; i < 16; ++i) sacc += acc0[i] + acc1[i];
    out[blockIdx.x * 256 + threadIdx.x] = x ^ __float_as_uint(sacc);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int T, int N, int MF>
double run(uint32_t* out, uint64_t* cyc, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<T, N, MF>), dim3(256), dim3(256), 0, 0, out, cyc, 64);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<T, N, MF>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    uint64_t c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("T=%2d N=%2d MF=%d: %8.1f ns / iter (wall), s_memtime ticks / iter %7.1f\n", T, N, MF, ms * 1e6 / iters, (double)c / iters);
    return ms;
}

#define ROW(T) run<T, 0, 1>(out, cyc, it); run<T, 4, 1>(out, cyc, it); run<T, 6, 1>(out, cyc, it); run<T, 8, 1>(out, cyc, it); run<T, 10, 1>(out, cyc, it); run<T, 12, 1>(out, cyc, it); run<T, 10, 0>(out, cyc, it);
int main2(uint32_t* out, uint64_t* cyc);
int main(int argc, char** argv) {
    uint32_t* out; uint64_t* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    if (argc > 1) return main2(out, cyc);
    const int it = 20000;
    const char* names[] = {"v_and_b32", "v_perm_b32", "v_pk_mul_f16", "v_pk_add_f16", "v_fma_mixlo_f16", "v_fma_f32", "v_and_or_b32", "v_pk_fma_f16",
                           "v_lshrrev_b32", "v_mul_f16", "v_cvt_f16_f32", "v_cvt_pkrtz_f16_f32", "v_mul_f32", "v_cvt_f32_ubyte0", "v_bfe_u32"};
    printf("%s\n", names[0]); ROW(0)
    printf("%s\n", names[1]); ROW(1)
    printf("%s\n", names[2]); ROW(2)
    printf("%s\n", names[3]); ROW(3)
    printf("%s\n", names[4]); ROW(4)
    printf("%s\n", names[5]); ROW(5)
    printf("%s\n", names[6]); ROW(6)
    printf("%s\n", names[7]); ROW(7)
    printf("%s\n", names[8]); ROW(8)
    printf("%s\n", names[9]); ROW(9)
    printf("%s\n", names[10]); ROW(10)
    printf("%s\n", names[11]); ROW(11)
    printf("%s\n", names[12]); ROW(12)
    printf("%s\n", names[13]); ROW(13)
    printf("%s\n", names[14]); ROW(14)
    return 0;
}

// ---- second question: TWO waves per SIMD with different roles.  Waves 0..3 issue only MFMAs (8 per iteration), waves 4..NW-1 only fillers
// (96 per iteration): do the two streams overlap on a SIMD (time = max) or add up?  cyc[0] = an MFMA wave's cycles, cyc[1] = a filler wave's.
template <int T, int NW>
__global__ __launch_bounds__(NW * 64) void roles(uint32_t* out, uint64_t* cyc, int iters) {
    const int wave = threadIdx.x >> 6;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f - i * 0.01f); }
    f16v acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    uint32_t r[16];
    for (int i = 0; i < 16; ++i) r[i] = threadIdx.x * 7 + i;
    uint32_t s0 = 0x3c003c00u + threadIdx.x, s1 = 0x0f0f0f0fu;
    float fs = 1.5f;
    uint64_t t0 = __builtin_readcyclecounter();
    if (wave < 4) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 96; ++i) filler<T>(r[i & 15], s0, s1, fs);
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    uint32_t x = 0;
    for (int i = 0; i < 16; ++i) x ^= r[i];
    float sacc = 0;
    for (int i = 0; i < 16; ++i) sacc += acc0[i] + acc1[i];
    out[blockIdx.x * NW * 64 + threadIdx.x] = x ^ __float_as_uint(sacc);
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    if (threadIdx.x == 256) cyc[1] = t1 - t0;
    if (threadIdx.x == 512) cyc[2] = t1 - t0;
}
template <int T, int NW>
void run_roles(uint32_t* out, uint64_t* cyc, int iters) {
    hipMemset(cyc, 0, 64);
    hipLaunchKernelGGL((roles<T, NW>), dim3(256), dim3(NW * 64), 0, 0, out, cyc, 64);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((roles<T, NW>), dim3(256), dim3(NW * 64), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    uint64_t c[3];
    hipMemcpy(c, cyc, 24, hipMemcpyDeviceToHost);
    printf("roles T=%2d waves/SIMD %d: MFMA wave %7.1f cycles / iter (8 MFMAs), filler wave %7.1f (96 fillers), second filler wave %7.1f\n", T, NW / 4,
           (double)c[0] / iters, (double)c[1] / iters, (double)c[2] / iters);
}
int main2(uint32_t* out, uint64_t* cyc) {
    const int it = 20000;
    run_roles<2, 4>(out, cyc, it);    // MFMA waves alone
    run_roles<2, 8>(out, cyc, it);    // + one v_pk_mul_f16 wave per SIMD
    run_roles<2, 12>(out, cyc, it);   // + two
    run_roles<1, 8>(out, cyc, it);    // v_perm_b32
    run_roles<1, 12>(out, cyc, it);
    run_roles<5, 8>(out, cyc, it);    // v_fma_f32
    run_roles<5, 12>(out, cyc, it);
    return 0;
}
