// Can an HBM-bound role and an MFMA-bound role share the CUs when they are blocks of ONE launch (interleaved by block index),
// with the resources of the heavier role (48 KiB LDS, ~160 VGPRs -> 3 blocks of 256 threads per CU)?
// Prints: streaming role alone, matrix role alone, both fused in one launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void stream_role(const uint4* __restrict__ src, float* __restrict__ out, int blk, size_t per_block_vec) {
    const uint4* p = src + (size_t)blk * per_block_vec;
    uint32_t acc = 0;
    for (size_t i = threadIdx.x; i < per_block_vec; i += 256 * 4) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (i + u * 256 < per_block_vec) ? p[i + u * 256] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[blk] = 1.f;  // keep the loads
}
__device__ __forceinline__ void matrix_role(float* __restrict__ out, int blk, int iters, char* smem) {
    f4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    h8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
        a[0] += (_Float16)0.001f;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0];
    if (s == 12345.f) out[blk] = s + smem[threadIdx.x];
}
// mode 0: all blocks stream; 1: all blocks matrix; 2: every P-th block is a matrix block
__global__ __launch_bounds__(256) void k(const uint4* src, float* out, size_t per_block_vec, int n_stream, int n_matrix, int iters, int mode, int P) {
    __shared__ char smem[49152];
    smem[threadIdx.x] = (char)threadIdx.x;
    int idx = blockIdx.x;
    if (mode == 0) { stream_role(src, out, idx, per_block_vec); return; }
    if (mode == 1) { matrix_role(out, idx, iters, smem); return; }
    const int grp = idx / P, pos = idx % P;
    if (pos == 0 && grp < n_matrix) matrix_role(out, grp, iters, smem);
    else {
        int sid = idx - (grp < n_matrix ? grp + 1 : n_matrix);
        if (sid < n_stream) stream_role(src, out, sid, per_block_vec);
    }
}
int main() {
    const size_t per_block = 160 * 1024, n_stream = 16384;  // 2.7 GB, like 512 requests x 32 heads x kv 512 of int8-g8 KV
    const int n_matrix = 768, iters = 120;                   // ~ a layer's GEMMs worth of MFMA time
    uint4* src; float* out;
    hipMalloc(&src, per_block * n_stream); hipMalloc(&out, (n_stream + n_matrix) * 4);
    hipMemset(src, 1, per_block * n_stream);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](int mode, int grid, int P) {
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, src, out, per_block / 16, (int)n_stream, n_matrix, iters, mode, P);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, src, out, per_block / 16, (int)n_stream, n_matrix, iters, mode, P);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        return ms / 5 * 1e3f;
    };
    const float ts = run(0, (int)n_stream, 1), tm = run(1, n_matrix, 1);
    printf("streaming role alone %.1f us (%.0f GB/s) | matrix role alone %.1f us | sum %.1f us\n", ts, per_block * n_stream / ts / 1e3, tm, ts + tm);
    for (int P : {4, 8, 16, 22}) {
        const int grid = (int)n_stream + n_matrix;
        printf("fused, every %2d-th block a matrix block: %.1f us\n", P, run(2, grid, P));
    }
    return 0;
}
