// Round 5 (go / no-go): what does a grid-wide hand-off INSIDE one launch cost on MI355X against the kernel boundary it
// would replace?  The seam of a batch-1 decode layer: 256 blocks each produce 16 values of a 4096-value row (a GEMV's outputs), then every
// block needs the whole row (the next GEMV's input).
//   two launches : kernel A writes the row, kernel B reads it (stream order is the hand-off)
//   one launch   : write-through stores (sc1), s_waitcnt, device-scope counter, bounded spin on the counter, L2-bypassing loads (sc1)
//   one launch, nothing exchanged : the same kernel without the seam (what the work itself costs)
// Each variant also streams `KB` KiB of weights per block per phase straight into registers (a stand-in for the GEMV's work), so that the
// numbers are seam costs of busy kernels, not of empty ones.   hipcc --offload-arch=gfx950 -O3 -o /tmp/gh grid_handoff_probe.hip; /tmp/gh
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int ROW = 4096, NB = 256, PER = ROW / NB;

__device__ __forceinline__ float stream_sum(const uint4* w, int n16, int tid) {   // n16 16-byte vectors per block, 256 threads
    float s = 0.f;
    for (int i = tid; i < n16; i += 256) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(w + i));
        s += __uint_as_float(((v[0] ^ v[1] ^ v[2] ^ v[3]) & 0x007fffffu) | 0x3f800000u);
    }
    return s;
}
__device__ __forceinline__ float block_sum(float v, float* red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(256) void phase_a(const uint4* w, int n16, float* row) {
    __shared__ float red[4];
    const float s = block_sum(stream_sum(w + (size_t)blockIdx.x * n16, n16, threadIdx.x), red);
    if (threadIdx.x < PER) row[blockIdx.x * PER + threadIdx.x] = s + threadIdx.x;
}
__global__ __launch_bounds__(256) void phase_b(const uint4* w, int n16, const float* row, float* out) {
    __shared__ float red[4];
    float x = 0.f;
    for (int i = threadIdx.x; i < ROW; i += 256) x += row[i];
    const float s = block_sum(stream_sum(w + (size_t)blockIdx.x * n16, n16, threadIdx.x) + x, red);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}
// one launch: mode 1 = with the seam, mode 2 = with the seam and phase B's first PF 16-byte vectors per thread loaded BEFORE the wait
// ("prefetch credit": the weight stream does not stop at the seam), mode 0 = without the seam (phase B reads a row written by an earlier launch)
__global__ __launch_bounds__(256) void fused(const uint4* wa, const uint4* wb, int n16, float* row, float* out, uint32_t* counter, uint32_t epoch,
                                             int mode, int* gave_up) {
    __shared__ float red[4];
    const float s = block_sum(stream_sum(wa + (size_t)blockIdx.x * n16, n16, threadIdx.x), red);
    constexpr int PF = 8;   // 8 x 16 B x 256 threads = 32 KiB of phase B's weights per block in registers across the seam
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 pf[PF];
    const uint4* wbb = wb + (size_t)blockIdx.x * n16;
    const int npf = mode == 2 ? (n16 / 256 < PF ? n16 / 256 : PF) : 0;
    if (mode == 2) {
#pragma unroll
        for (int i = 0; i < PF; ++i)
            if (i < npf) pf[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wbb + threadIdx.x + i * 256));
    }
    if (mode) {
        if (threadIdx.x < PER) {
            float* p = row + blockIdx.x * PER + threadIdx.x;
            asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(s + threadIdx.x) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * NB) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 2000000) { *gave_up = 1; break; }
            }
        }
        __syncthreads();
    } else if (threadIdx.x < PER) {
        row[ROW + blockIdx.x * PER + threadIdx.x] = s + threadIdx.x;   // (same store traffic, nobody reads it)
    }
    float x = 0.f;
    if (mode) {   // 16 values per thread: four 16-byte loads past the L2, ONE wait
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        f32x4 a, b, c, d;
        const float* p = row + threadIdx.x * 4;
        asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
                     "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p), "v"(p + 1024), "v"(p + 2048), "v"(p + 3072) : "memory");
        x = (a[0] + a[1] + a[2] + a[3]) + (b[0] + b[1] + b[2] + b[3]) + (c[0] + c[1] + c[2] + c[3]) + (d[0] + d[1] + d[2] + d[3]);
    } else {
        for (int i = threadIdx.x; i < ROW; i += 256) x += row[i];
    }
    float acc = 0.f;
    if (mode == 2) {
#pragma unroll
        for (int i = 0; i < PF; ++i)
            if (i < npf) acc += __uint_as_float(((pf[i][0] ^ pf[i][1] ^ pf[i][2] ^ pf[i][3]) & 0x007fffffu) | 0x3f800000u);
    }
    const float t = block_sum(stream_sum(wbb + npf * 256, n16 - npf * 256, threadIdx.x) + acc + x, red);
    if (threadIdx.x == 0) out[blockIdx.x] = t;
}

int main() {
    const int reps = 400;
    for (int KB : {0, 16, 64, 256}) {
        const int n16 = KB * 1024 / 16;
        // a ring of weight copies larger than the Infinity Cache (every launch streams from HBM)
        const size_t per = (size_t)NB * n16 * 16 * 2;   // both phases
        const int copies = per ? (int)(600e6 / per) + 2 : 1;
        uint4* w; float *row, *out; uint32_t* counter; int* gave_up;
        CK(hipMalloc(&w, per * copies + 64)); CK(hipMemset(w, 1, per * copies + 64));
        CK(hipMalloc(&row, 2 * ROW * 4)); CK(hipMalloc(&out, NB * 4)); CK(hipMalloc(&counter, 4)); CK(hipMalloc(&gave_up, 4));
        CK(hipMemset(counter, 0, 4)); CK(hipMemset(gave_up, 0, 4)); CK(hipMemset(row, 0, 2 * ROW * 4));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float ms[4];
        uint32_t epoch = 0;
        for (int variant = 0; variant < 4; ++variant) {
            for (int pass = 0; pass < 2; ++pass) {   // warm, then timed
                CK(hipEventRecord(e0));
                for (int r = 0; r < reps; ++r) {
                    const uint4* wa = w + (size_t)(r % copies) * (per / 16);
                    const uint4* wb = wa + (size_t)NB * n16;
                    if (variant == 0) {
                        hipLaunchKernelGGL(phase_a, dim3(NB), dim3(256), 0, 0, wa, n16, row);
                        hipLaunchKernelGGL(phase_b, dim3(NB), dim3(256), 0, 0, wb, n16, row, out);
                    } else {
                        if (variant == 1 || variant == 3) ++epoch;
                        hipLaunchKernelGGL(fused, dim3(NB), dim3(256), 0, 0, wa, wb, n16, row, out, counter, epoch, variant == 1 ? 1 : (variant == 3 ? 2 : 0), gave_up);
                    }
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms[variant], e0, e1));
            }
        }
        int g = 0; CK(hipMemcpy(&g, gave_up, 4, hipMemcpyDeviceToHost));
        printf("%4d KiB per block and phase (%5.1f MB per phase): two launches %6.2f us | one launch with the hand-off %6.2f us | one launch, no hand-off %6.2f us | hand-off with 32 KiB per block prefetched across it %6.2f us | gave up %d\n",
               KB, NB * KB / 1024.0, ms[0] / reps * 1e3, ms[1] / reps * 1e3, ms[2] / reps * 1e3, ms[3] / reps * 1e3, g);
        CK(hipFree(w)); CK(hipFree(row)); CK(hipFree(out)); CK(hipFree(counter)); CK(hipFree(gave_up));
    }
    return 0;
}
