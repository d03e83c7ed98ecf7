// Tensor-parallel collectives written for the MI355X node itself: 8 GPUs, every pair joined by its own xGMI link
// (7 links x ~153 GB/s per GPU, no switch).  A ring all-reduce moves 2(N-1)/N of the message over ONE link per hop and is
// bound by that link; here every rank talks to all its peers at once, so a 8 MiB all-reduce at N = 8 puts 1 MiB on each
// of the 7 links per phase (~14 us against ~96 us for a ring).
//
//   all-reduce(sum) of fp16[count], in place in a SYMMETRIC buffer (same offset in every rank's exchange region):
//     1. start barrier  : block b of every rank tells block b of every peer "my partial sums are complete" (the kernel is
//                         stream-ordered after the GEMM that produced them) and waits for the same from all of them;
//     2. reduce-scatter : rank r owns the r-th 1/N of the buffer: it PULLS that slice from every rank (peer reads over the
//                         rank's N-1 links in parallel), adds in fp32 in rank order 0..N-1 -- one rounding to fp16, the
//                         oracle's arithmetic (oracle/llama_ref.c ref_forward) -- and
//     3. all-gather     : PUSHES the reduced slice into the same place of every rank's buffer (peer writes);
//     4. end barrier    : block b waits until block b of every peer has finished writing.
//   all-gather of the fp32 logits shards: every rank pushes its [B, V/N] block into slot `me` of every peer's gather
//   buffer between a start and an end barrier.
//
// Visibility rules this relies on (nothing else): the exchange region is allocated UNCACHED (hipDeviceMallocUncached: no
// L2 residency on either side); every access a peer must see or that reads a peer's data is a SYSTEM-scope relaxed atomic
// (global_load/store ... sc0 sc1: misses / writes through every cache level); every writing wave drains its stores
// (s_waitcnt vmcnt(0)) before the block's flag is raised; flags are 32-bit epochs compared with wrap-safe >=, one word per
// (block, source rank), so nothing is ever reset and a late reader of epoch e is not confused by e + 1.
// All spins are bounded (s_memrealtime): on a timeout the kernel raises a status word in host memory and returns instead of
// hanging the GPU; the runtime turns that into an error at the step's synchronisation point.
#include "kernels.h"

namespace pplhip {

namespace {

typedef unsigned long long u64;

__device__ __forceinline__ uint32_t ld_sys(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_sys(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ u64 ld_sys(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_sys(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// block b of this rank <-> block b of every peer.  END: the block's data stores must have landed before the flag goes out.
template <bool END>
__device__ __forceinline__ void cross_rank_barrier(const P2pPeers& peers, int me, int n, size_t flag_off, uint32_t epoch,
                                                   u64 timeout_ticks, uint32_t* status) {
    if (END) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every writing wave drains its own stores
        __syncthreads();
    }
    const int t = threadIdx.x;
    if (t < n && t != me) {
        if (END) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        uint32_t* theirs = reinterpret_cast<uint32_t*>(peers.base[t] + flag_off) + (size_t)blockIdx.x * P2P_MAX_RANKS + me;
        st_sys(theirs, epoch);
        const uint32_t* mine = reinterpret_cast<const uint32_t*>(peers.base[me] + flag_off) + (size_t)blockIdx.x * P2P_MAX_RANKS + t;
        const u64 t0 = wall_clock64();
        while ((int32_t)(ld_sys(mine) - epoch) < 0) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > timeout_ticks) {
                __hip_atomic_store(status, 1u + (uint32_t)t + (END ? 16u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
    __syncthreads();
}

__device__ __forceinline__ void add4(float* acc, u64 v) {
    const h4 h = __builtin_bit_cast(h4, v);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] += (float)h[i];
}

// N > 0: compile-time rank count (2, 4, 8); N == 0: any n <= P2P_MAX_RANKS
template <int N>
__global__ __launch_bounds__(512) void p2p_allreduce_kernel(P2pPeers peers, int me, int n_rt, size_t data_off, int64_t granules,
                                                            uint32_t epoch, u64 timeout_ticks, uint32_t* status) {
    const int n = N > 0 ? N : n_rt;
    cross_rank_barrier<false>(peers, me, n, P2P_FLAGS_START, epoch, timeout_ticks, status);
    const int64_t per = (granules + n - 1) / n;
    const int64_t lo = per * me, hi = (lo + per < granules) ? lo + per : granules;
    constexpr int MAXN = N > 0 ? N : P2P_MAX_RANKS;
    for (int64_t g = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < hi; g += (int64_t)gridDim.x * blockDim.x) {
        u64 v[MAXN];
#pragma unroll
        for (int r = 0; r < MAXN; ++r)
            if (r < n) v[r] = ld_sys(reinterpret_cast<const u64*>(peers.base[r] + data_off) + g);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < MAXN; ++r)
            if (r < n) add4(acc, v[r]);  // rank order 0 .. n-1 on every element: all ranks hold the same bits afterwards
        const h4 o = {to_h(acc[0]), to_h(acc[1]), to_h(acc[2]), to_h(acc[3])};
        const u64 ov = __builtin_bit_cast(u64, o);
#pragma unroll
        for (int r = 0; r < MAXN; ++r)
            if (r < n) st_sys(reinterpret_cast<u64*>(peers.base[r] + data_off) + g, ov);
    }
    cross_rank_barrier<true>(peers, me, n, P2P_FLAGS_END, epoch, timeout_ticks, status);
}

// every rank pushes src[granules] (its own, ordinary memory) into slot `me` (slot_granules apart) of every rank's gather buffer
__global__ __launch_bounds__(512) void p2p_allgather_kernel(P2pPeers peers, int me, int n, const u64* __restrict__ src, size_t dst_off,
                                                            int64_t granules, int64_t slot_granules, uint32_t epoch,
                                                            u64 timeout_ticks, uint32_t* status) {
    cross_rank_barrier<false>(peers, me, n, P2P_FLAGS_START, epoch, timeout_ticks, status);
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < granules; g += (int64_t)gridDim.x * blockDim.x) {
        const u64 v = src[g];
        for (int r = 0; r < n; ++r) st_sys(reinterpret_cast<u64*>(peers.base[r] + dst_off) + (int64_t)me * slot_granules + g, v);
    }
    cross_rank_barrier<true>(peers, me, n, P2P_FLAGS_END, epoch, timeout_ticks, status);
}

int grid_for(int64_t granules_per_rank) {
    // 512 threads x up to 8 granules each per block; enough blocks to keep every link busy, few enough that the spinning
    // blocks of several ranks emulated on ONE device (tests) never fill it
    int64_t b = (granules_per_rank + 512 * 8 - 1) / (512 * 8);
    if (b < 1) b = 1;
    static const int cap = getenv("PPLHIP_P2P_BLOCKS") ? atoi(getenv("PPLHIP_P2P_BLOCKS")) : 32;
    const int c = cap < 1 ? 1 : (cap > P2P_MAX_BLOCKS ? P2P_MAX_BLOCKS : cap);
    return (int)(b > c ? c : b);
}

}  // namespace

hipError_t launch_p2p_allreduce(hipStream_t s, const P2pPeers& peers, int me, int n, size_t data_off, int64_t count, uint32_t epoch,
                                uint64_t timeout_ticks, uint32_t* status) {
    if (n < 2 || n > P2P_MAX_RANKS || count % 4 || data_off % 8) return hipErrorInvalidValue;
    if (count == 0) return hipSuccess;
    const int64_t granules = count / 4;
    const dim3 grid(grid_for((granules + n - 1) / n)), block(512);
    if (n == 2) hipLaunchKernelGGL(p2p_allreduce_kernel<2>, grid, block, 0, s, peers, me, n, data_off, granules, epoch, (u64)timeout_ticks, status);
    else if (n == 4) hipLaunchKernelGGL(p2p_allreduce_kernel<4>, grid, block, 0, s, peers, me, n, data_off, granules, epoch, (u64)timeout_ticks, status);
    else if (n == 8) hipLaunchKernelGGL(p2p_allreduce_kernel<8>, grid, block, 0, s, peers, me, n, data_off, granules, epoch, (u64)timeout_ticks, status);
    else hipLaunchKernelGGL(p2p_allreduce_kernel<0>, grid, block, 0, s, peers, me, n, data_off, granules, epoch, (u64)timeout_ticks, status);
    return hipGetLastError();
}

hipError_t launch_p2p_allgather(hipStream_t s, const P2pPeers& peers, int me, int n, const void* src, size_t dst_off, int64_t bytes,
                                int64_t slot_bytes, uint32_t epoch, uint64_t timeout_ticks, uint32_t* status) {
    if (n < 2 || n > P2P_MAX_RANKS || bytes % 8 || slot_bytes % 8 || dst_off % 8) return hipErrorInvalidValue;
    if (bytes == 0) return hipSuccess;
    const dim3 grid(grid_for(bytes / 8)), block(512);
    hipLaunchKernelGGL(p2p_allgather_kernel, grid, block, 0, s, peers, me, n, reinterpret_cast<const u64*>(src), dst_off, bytes / 8,
                       slot_bytes / 8, epoch, (u64)timeout_ticks, status);
    return hipGetLastError();
}

}  // namespace pplhip
