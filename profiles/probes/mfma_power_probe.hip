// Pure-MFMA streams on random vs zero operands: does the power limit treat v_mfma_f32_16x16x32_f16 and v_mfma_f32_32x32x16_f16 alike?
// (round 4: the hand-scheduled 32x32x16 loop and the compiler-scheduled 16x16x32 kernel land on the same rate at different clocks.)
// build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC mfma_power_probe.hip -o libmfma_power_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int SHAPE>  // 16 or 32
__global__ __launch_bounds__(256) void mfma_stream(const uint4* __restrict__ a_in, const uint4* __restrict__ b_in, float* out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    h8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(h8, a_in[(tid * 4 + i) & 0xffff]);
        b[i] = __builtin_bit_cast(h8, b_in[(tid * 4 + i) & 0xffff]);
    }
    float s = 0.f;
    if constexpr (SHAPE == 16) {
        f4 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    } else {
        f16v acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
    }
    if (s == 123.456f) out[tid] = s;
}

extern "C" int mfma_probe(int shape, int blocks, int iters, const void* a, const void* b, void* out, void* stream) {
    if (shape == 16) hipLaunchKernelGGL(mfma_stream<16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)a, (const uint4*)b, (float*)out, iters);
    else hipLaunchKernelGGL(mfma_stream<32>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)a, (const uint4*)b, (float*)out, iters);
    return (int)hipGetLastError();
}
