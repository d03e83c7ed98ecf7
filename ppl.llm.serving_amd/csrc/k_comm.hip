// Tensor-parallel collectives written for the MI355X node itself: 8 GPUs, every pair joined by its own xGMI link
// (7 links x ~153 GB/s per GPU, no switch).  A ring all-reduce moves 2(N-1)/N of the message over ONE link per hop and is
// bound by that link; here every rank talks to all its peers at once, so a 8 MiB all-reduce at N = 8 puts 1 MiB on each
// of the 7 links per phase (~14 us against ~96 us for a ring).
//
//   all-reduce(sum) of fp16[count] held at the same offset of every rank's exchange region, two-shot, PULL only:
//     1. start barrier  : block b of every rank tells block b of every peer "my partial sums are complete" (the kernel is
//                         stream-ordered after the GEMM that produced them) and waits for the same from all of them;
//     2. reduce-scatter : rank r owns the r-th 1/N of the buffer: it reads that slice of every rank (peer reads over the
//                         rank's N-1 links in parallel), adds in fp32 in rank order 0..N-1 -- one rounding to fp16, the
//                         oracle's arithmetic (oracle/llama_ref.c ref_forward) -- and stores the result in its OWN scratch
//                         slice (exchange region, double-buffered by epoch parity);
//     3. middle barrier : "my reduced slice is published, and I am done reading your partial sums";
//     4. all-gather     : every rank reads the N reduced slices from their owners' scratch and writes the whole result over
//                         its own buffer with ordinary stores -- a kernel's plain output, consumed by the next kernel like
//                         any other.  No end barrier: a rank's scratch is rewritten two collectives later, behind two more
//                         start barriers that every peer only passes after leaving this kernel.
//   Peers never WRITE each other's data, only flag words: a remote store into memory that a later local kernel reads with
//   ordinary loads would depend on that GPU's caches holding no older copy of the line.
//   all-gather of the fp32 logits shards: start barrier, every rank reads every peer's shard (published in the exchange
//   region by a local copy) into its own gather buffer.
//
// Visibility rules this relies on (nothing else): the exchange region is FINE-GRAINED device memory (coherent between
// devices at system scope; pplhip.cc explains why not "uncached");
// every access that reads a peer's data or publishes data for a peer is a SYSTEM-scope relaxed atomic (global_load/store
// ... sc0 sc1: misses / writes through every cache level); every wave that published data drains its stores (s_waitcnt
// vmcnt(0)) before the block's flag is raised; flags are 32-bit epochs compared with wrap-safe >=, one word per (barrier
// kind, block, source rank), so nothing is ever reset and a late reader of epoch e is not confused by e + 1.
// All spins are bounded (s_memrealtime): on a timeout the kernel raises a status word in host memory and returns instead of
// hanging the GPU; the runtime turns that into an error at the step's synchronisation point.
#include "kernels.h"

namespace pplhip {

namespace {

typedef unsigned long long u64;

__device__ __forceinline__ uint32_t ld_sys(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_sys(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ u64 ld_sys(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// 16-byte system-scope accesses (global_load/store_dwordx4 ... sc0 sc1): the same cache-bypass bits the 8-byte atomics above
// compile to, at the widest access the memory pipeline has -- half the instructions per byte pulled over a link.  Issued from
// inline asm (HIP has no 16-byte atomic): the compiler does not count them, so a batch of loads and its s_waitcnt vmcnt(0) are one
// asm statement (ld_sys16_batch).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_sys16(void* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
// A batch of N 16-byte system-scope loads AND their s_waitcnt in ONE asm statement (early-clobber outputs): the compiler never sees a
// destination register between its load and the wait, so no copy / phi / spill of it can read stale data (the loads used
// to be separate statements tied to a later wait only through register constraints).  Every load is unconditional -- callers clamp the
// address of a lane that has nothing to fetch to a valid one and ignore the result.
#define LD1 "global_load_dwordx4 %0, %2, off sc0 sc1\n\tglobal_load_dwordx4 %1, %3, off sc0 sc1\n\t"
__device__ __forceinline__ void ld_sys16x2(u32x4& a, u32x4& b, const void* pa, const void* pb) {
    asm volatile(LD1 "s_waitcnt vmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(pa), "v"(pb) : "memory");
}
#undef LD1
__device__ __forceinline__ void ld_sys16x4(u32x4& a, u32x4& b, u32x4& c, u32x4& d, const void* pa, const void* pb, const void* pc, const void* pd) {
    asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\tglobal_load_dwordx4 %1, %5, off sc0 sc1\n\t"
                 "global_load_dwordx4 %2, %6, off sc0 sc1\n\tglobal_load_dwordx4 %3, %7, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(pa), "v"(pb), "v"(pc), "v"(pd) : "memory");
}
template <int N>
__device__ __forceinline__ void ld_sys16_batch(u32x4 (&v)[N], const void* (&p)[N]) {
    static_assert(N == 2 || N == 4 || N == 8, "rank counts 2 / 4 / 8 (the generic kernel pads to 8)");
    if constexpr (N == 2) {
        ld_sys16x2(v[0], v[1], p[0], p[1]);
    } else if constexpr (N == 4) {
        ld_sys16x4(v[0], v[1], v[2], v[3], p[0], p[1], p[2], p[3]);
    } else {
        asm volatile("global_load_dwordx4 %0, %8, off sc0 sc1\n\tglobal_load_dwordx4 %1, %9, off sc0 sc1\n\t"
                     "global_load_dwordx4 %2, %10, off sc0 sc1\n\tglobal_load_dwordx4 %3, %11, off sc0 sc1\n\t"
                     "global_load_dwordx4 %4, %12, off sc0 sc1\n\tglobal_load_dwordx4 %5, %13, off sc0 sc1\n\t"
                     "global_load_dwordx4 %6, %14, off sc0 sc1\n\tglobal_load_dwordx4 %7, %15, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                     : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])
                     : "memory");
    }
}

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

__device__ __forceinline__ void add8(float* acc, u32x4 v) {
    const h8 h = __builtin_bit_cast(h8, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += (float)h[i];
}

// N > 0: compile-time rank count (2, 4, 8); N == 0: any n <= P2P_MAX_RANKS
template <int N>
__global__ __launch_bounds__(512) void p2p_allreduce_kernel(P2pPeers peers, int me, int n_rt, size_t data_off, size_t scratch_off,
                                                            int64_t granules, uint32_t epoch, u64 timeout_ticks, uint32_t* status,
                                                            size_t flags_off /* channel: a flag set of its own */) {
    const int n = N > 0 ? N : n_rt;
    constexpr int MAXN = N > 0 ? N : P2P_MAX_RANKS;
    const int64_t per = (granules + n - 1) / n;  // slice of rank r: granules [r * per, min((r + 1) * per, granules))
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (int64_t)gridDim.x * blockDim.x;
    cross_rank_barrier<false>(peers, me, n, P2P_FLAGS_START + flags_off, epoch, timeout_ticks, status);
    {   // reduce-scatter into my scratch slice (granule = 16 bytes = 8 fp16)
        const int64_t lo = per * me, hi = (lo + per < granules) ? lo + per : granules;
        char* mine = peers.base[me] + scratch_off;
        for (int64_t g = lo + tid; g < hi; g += nthr) {
            u32x4 v[MAXN];
            const void* src[MAXN];
#pragma unroll
            for (int r = 0; r < MAXN; ++r) src[r] = peers.base[r < n ? r : me] + data_off + g * 16;  // (a rank slot past n re-reads my own copy)
            ld_sys16_batch<MAXN>(v, src);
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < MAXN; ++r)
                if (r < n) add8(acc, v[r]);  // rank order 0 .. n-1 on every element: one result for the whole group
            const h8 o = {to_h(acc[0]), to_h(acc[1]), to_h(acc[2]), to_h(acc[3]), to_h(acc[4]), to_h(acc[5]), to_h(acc[6]), to_h(acc[7])};
            st_sys16(mine + (g - lo) * 16, __builtin_bit_cast(u32x4, o));
        }
    }
    cross_rank_barrier<true>(peers, me, n, P2P_FLAGS_MID + flags_off, epoch, timeout_ticks, status);
    {   // all-gather: slice r from rank r's scratch -> my buffer (ordinary stores)
        u32x4* out = reinterpret_cast<u32x4*>(peers.base[me] + data_off);
        for (int64_t g = tid; g < per; g += nthr) {
            u32x4 v[MAXN];
            const void* src[MAXN];
#pragma unroll
            for (int r = 0; r < MAXN; ++r)  // (granules past the end of the last slice: scratch holds room for `per` granules on every rank; not stored)
                src[r] = peers.base[r < n ? r : me] + scratch_off + g * 16;
            ld_sys16_batch<MAXN>(v, src);
#pragma unroll
            for (int r = 0; r < MAXN; ++r)
                if (r < n && per * r + g < granules) out[per * r + g] = v[r];
        }
    }
}

// ---- all-reduce fused with the (Skip)RMSNorm that consumes it: the sequence-parallel residual stream (round 6) ------------------------
// The two-shot all-reduce above leaves rank r, after its first shot, with the reduced sums of ITS 1/N of the buffer -- and then every rank
// gathers all N slices and runs the SAME residual-add + RMSNorm over all T rows: N-fold replicated work and one more launch per
// half-layer (7B at tensor-parallel 8, 1024 rows: 2 x 7.4 us of 215 us per layer and rank, profiles/r06_tp8_slice_kernel_stats.csv).
// Here the slices are whole ROWS (rank r owns rows [r per, (r + 1) per), per = ceil(rows / N)) and the owner does the rest of the
// half-layer's element-wise work on them between the shots:
//     1. start barrier
//     2. for each owned row: pull the N partial-sum rows (peer reads over the N - 1 links at once), s = fp16(sum in rank order) -- the
//        all-reduce's arithmetic; r = fp16(h + s) -> h (the residual stream: each rank keeps ONLY its own rows of it from here on);
//        y = fp16(r * rsqrt(mean(r^2) + eps) * w) -> the local normed matrix AND the rank's scratch slice (system-scope stores)
//     3. middle barrier
//     4. gather: every rank pulls the NORMED rows of the other owners into its local matrix (ordinary stores)
// The same bytes cross the links as before (reduce-scatter + all-gather of fp16 [rows, hidden]); the norm runs on 1/N of the rows; the
// next GEMM starts from the gathered matrix.  A block handles whole rows (the norm is a row reduction) and block b gathers exactly the
// rows block b of the owner produced, so the per-block cross-rank barriers of the plain kernel still order everything.
// N = 1 ("solo"): no peers, no barriers, every row owned -- the self-test's local reference (same arithmetic, bit for bit).
template <int N>
__global__ __launch_bounds__(512) void p2p_allreduce_norm_kernel(P2pPeers peers, int me, int n_rt, size_t data_off, size_t scratch_off,
                                                                 int64_t rows, int chunks /* hidden / 8, <= 1024 */, int hidden,
                                                                 uint4* h, const uint4* __restrict__ w, float eps, uint4* __restrict__ xn,
                                                                 uint32_t epoch, u64 timeout_ticks, uint32_t* status, size_t flags_off) {
    constexpr int MC = 2;                                  // chunks (8 fp16) of a row per thread
    const int n = N > 0 ? N : n_rt;
    constexpr int MAXN = N > 0 ? N : P2P_MAX_RANKS;
    __shared__ float red[8];
    const int64_t per = (rows + n - 1) / n;
    const int64_t lo = N == 1 ? 0 : per * me, hi = (lo + per < rows) ? lo + per : rows;   // (solo: every row, whatever the rank's number)
    const int t = threadIdx.x;
    if (N != 1) cross_rank_barrier<false>(peers, me, n, P2P_FLAGS_START + flags_off, epoch, timeout_ticks, status);
    char* mine = peers.base[me] + scratch_off;
    for (int64_t row = lo + blockIdx.x; row < hi; row += gridDim.x) {
        float v[MC][8];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < MC; ++i) {
            const int c = t + i * 512;
            if (i * 512 < chunks) {                        // (block-uniform: the batched loads below are unconditional per lane)
                const int cc = c < chunks ? c : chunks - 1;
                float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if constexpr (N == 1) {
                    unpack8(*reinterpret_cast<const uint4*>(peers.base[me] + data_off + (row * chunks + cc) * 16), acc);
                } else {
                    u32x4 p[MAXN];
                    const void* src[MAXN];
#pragma unroll
                    for (int r = 0; r < MAXN; ++r) src[r] = peers.base[r < n ? r : me] + data_off + (row * chunks + cc) * 16;
                    ld_sys16_batch<MAXN>(p, src);
#pragma unroll
                    for (int r = 0; r < MAXN; ++r)
                        if (r < n) add8(acc, p[r]);        // rank order 0 .. n-1: p2p_allreduce_kernel's sum
                }
                float hx[8];
                unpack8(h[row * chunks + cc], hx);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[i][j] = round_h(hx[j] + round_h(acc[j]));   // fp16(h + fp16(sum)): all-reduce, then SkipRMSNorm's add
                if (c < chunks) {
                    h[row * chunks + c] = pack8(v[i]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) ss += v[i][j] * v[i][j];
                }
            }
        }
        uint4 wraw[MC];
#pragma unroll
        for (int i = 0; i < MC; ++i) {
            const int c = t + i * 512;
            wraw[i] = c < chunks ? w[c] : make_uint4(0, 0, 0, 0);
        }
        ss = wave_sum(ss);
        if ((t & 63) == 0) red[t >> 6] = ss;
        __syncthreads();
        ss = red[0] + red[1] + red[2] + red[3];
#pragma unroll
        for (int wv = 4; wv < 8; ++wv) ss += red[wv];
        const float inv = 1.0f / sqrtf(ss / (float)hidden + eps);
#pragma unroll
        for (int i = 0; i < MC; ++i) {
            const int c = t + i * 512;
            if (c < chunks) {
                float wf[8], o[8];
                unpack8(wraw[i], wf);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = v[i][j] * inv * wf[j];
                const uint4 y = pack8(o);
                xn[row * chunks + c] = y;
                if (N != 1) st_sys16(mine + ((row - lo) * chunks + c) * 16, __builtin_bit_cast(u32x4, y));
            }
        }
        __syncthreads();                                   // red is reused by the next row
    }
    if constexpr (N != 1) {
    cross_rank_barrier<true>(peers, me, n, P2P_FLAGS_MID + flags_off, epoch, timeout_ticks, status);
    for (int64_t j = blockIdx.x; j < per; j += gridDim.x) {   // local row j of every owner: written there by ITS block blockIdx.x
#pragma unroll
        for (int i = 0; i < MC; ++i) {
            const int c = t + i * 512;
            if (i * 512 < chunks) {
                const int cc = c < chunks ? c : chunks - 1;
                u32x4 p[MAXN];
                const void* src[MAXN];
#pragma unroll
                for (int r = 0; r < MAXN; ++r) src[r] = peers.base[r < n ? r : me] + scratch_off + (j * chunks + cc) * 16;
                ld_sys16_batch<MAXN>(p, src);
#pragma unroll
                for (int r = 0; r < MAXN; ++r)
                    if (r < n && r != me && c < chunks && per * r + j < rows) xn[(per * r + j) * chunks + c] = __builtin_bit_cast(uint4, p[r]);
            }
        }
    }
    }
}

// logits shards: rows x row_granules granules (G = 8 or 16 bytes) at src_off of every rank's region (rank r's [rows, V/n] block)
// -> columns [r * row_granules, (r + 1) * row_granules) of my [rows, dst_row_granules] matrix (ordinary local memory)
template <int GB>
__global__ __launch_bounds__(512) void p2p_allgather_kernel(P2pPeers peers, int me, int n, size_t src_off, char* __restrict__ dst,
                                                            int64_t rows, int64_t row_granules, int64_t dst_row_granules,
                                                            uint32_t epoch, u64 timeout_ticks, uint32_t* status) {
    cross_rank_barrier<false>(peers, me, n, P2P_FLAGS_START, epoch, timeout_ticks, status);
    const int64_t total = rows * row_granules;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = g / row_granules, col = g - row * row_granules;
        if (GB == 16) {
            u32x4 v[P2P_MAX_RANKS];
            const void* src[P2P_MAX_RANKS];
#pragma unroll
            for (int r = 0; r < P2P_MAX_RANKS; ++r) src[r] = peers.base[r < n ? r : me] + src_off + g * 16;
            ld_sys16_batch<P2P_MAX_RANKS>(v, src);
#pragma unroll
            for (int r = 0; r < P2P_MAX_RANKS; ++r)
                if (r < n) reinterpret_cast<u32x4*>(dst)[row * dst_row_granules + (int64_t)r * row_granules + col] = v[r];
        } else {
            for (int r = 0; r < n; ++r)
                reinterpret_cast<u64*>(dst)[row * dst_row_granules + (int64_t)r * row_granules + col] =
                    ld_sys(reinterpret_cast<const u64*>(peers.base[r] + src_off) + g);
        }
    }
}

// ---- stream hand-off through device memory -----------------------------------------------------------------------------------
// The two-chunk schedule passes work between the rank's compute stream and its communication stream four times per layer.  As HIP
// events (hipEventRecord + hipStreamWaitEvent) every hand-off is a barrier packet the command processor has to retire before the
// other queue moves: measured ~18 us per dependency, 4.8 ms per 1024-row decode step at tp 8 -- more than the collectives it hides.
// Here the dependency is a 32-bit epoch in device memory: the producing stream raises it with a one-thread kernel AFTER the kernel
// whose output it publishes (stream order), the consuming stream's next kernel is preceded by a one-wave kernel that spins on it
// (bounded like the collectives' barriers).  Kernel boundaries on either side give the release / acquire at agent scope.
__global__ void handoff_signal_kernel(uint32_t* flag, uint32_t epoch) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// waits for *wait_flag >= wait_epoch (wrap-safe), then (optionally) raises *done_flag: with both it is the whole "identity
// collective" of a tensor-parallel step whose ranks are emulated (bench.py --emulate-tp)
__global__ void handoff_wait_kernel(const uint32_t* wait_flag, uint32_t wait_epoch, uint32_t* done_flag, uint32_t done_epoch,
                                    u64 timeout_ticks, uint32_t* status) {
    if (threadIdx.x == 0) {
        const u64 t0 = wall_clock64();
        while ((int32_t)(__hip_atomic_load(wait_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - wait_epoch) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > timeout_ticks) {
                __hip_atomic_store(status, 64u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (done_flag) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_store(done_flag, done_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// self-test inputs, produced the way the real inputs are: by a kernel on the rank's stream (exactly representable values
// whose sums over <= 8 ranks are exact in fp16)
__device__ __forceinline__ float p2p_pat(int64_t i, int g, int round) { return (float)((int)((i * 7 + g * 13 + round * 5) % 64) - 32) * 0.25f; }
__global__ __launch_bounds__(256) void p2p_pattern_kernel(uint16_t* __restrict__ halfs, int64_t cnt, float* __restrict__ floats, int64_t gcnt,
                                                          int g, int round) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * 256) halfs[i] = f2h(p2p_pat(i, g, round));
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < gcnt; i += (int64_t)gridDim.x * 256) floats[i] = p2p_pat(i, g, round + 2);
}

int grid_for(int64_t granules_per_rank) {
    // 512 threads x up to 8 granules (16 B) each per block; enough blocks to keep every link busy, few enough that the spinning
    // blocks of several ranks emulated on ONE device (tests) never fill it
    int64_t b = (granules_per_rank + 512 * 8 - 1) / (512 * 8);
    if (b < 1) b = 1;
    static const int cap = tune_int("PPLHIP_P2P_BLOCKS", 32);
    const int c = cap < 1 ? 1 : (cap > P2P_MAX_BLOCKS ? P2P_MAX_BLOCKS : cap);
    return (int)(b > c ? c : b);
}

}  // namespace

hipError_t launch_handoff_signal(hipStream_t s, uint32_t* flag, uint32_t epoch) {
    hipLaunchKernelGGL(handoff_signal_kernel, dim3(1), dim3(1), 0, s, flag, epoch);
    return hipGetLastError();
}
hipError_t launch_handoff_wait(hipStream_t s, const uint32_t* wait_flag, uint32_t wait_epoch, uint32_t* done_flag, uint32_t done_epoch,
                               uint64_t timeout_ticks, uint32_t* status) {
    hipLaunchKernelGGL(handoff_wait_kernel, dim3(1), dim3(64), 0, s, wait_flag, wait_epoch, done_flag, done_epoch, (u64)timeout_ticks, status);
    return hipGetLastError();
}

float p2p_pattern_value(int64_t i, int g, int round) { return (float)((int)((i * 7 + g * 13 + round * 5) % 64) - 32) * 0.25f; }

hipError_t launch_p2p_pattern(hipStream_t s, uint16_t* halfs, int64_t cnt, float* floats, int64_t gcnt, int g, int round) {
    hipLaunchKernelGGL(p2p_pattern_kernel, dim3(256), dim3(256), 0, s, halfs, cnt, floats, gcnt, g, round);
    return hipGetLastError();
}

hipError_t launch_p2p_allreduce(hipStream_t s, const P2pPeers& peers, int me, int n, size_t data_off, size_t scratch_off, int64_t count,
                                uint32_t epoch, uint64_t timeout_ticks, uint32_t* status, int channel) {
    if (n < 2 || n > P2P_MAX_RANKS || count % 8 || data_off % 16 || scratch_off % 16 || channel < 0 || channel >= P2P_CHANNELS) return hipErrorInvalidValue;
    const size_t flags_off = (size_t)channel * P2P_CHANNEL_FLAG_BYTES;
    if (count == 0) return hipSuccess;
    const int64_t granules = count / 8;
    const dim3 grid(grid_for((granules + n - 1) / n)), block(512);
#define AR(NN) hipLaunchKernelGGL(p2p_allreduce_kernel<NN>, grid, block, 0, s, peers, me, n, data_off, scratch_off, granules, epoch, (u64)timeout_ticks, status, flags_off)
    if (n == 2) AR(2); else if (n == 4) AR(4); else if (n == 8) AR(8); else AR(0);
#undef AR
    return hipGetLastError();
}

hipError_t launch_p2p_allreduce_norm(hipStream_t s, const P2pPeers& peers, int me, int n, size_t data_off, size_t scratch_off, int64_t rows,
                                     int hidden, uint16_t* h, const uint16_t* w, float eps, uint16_t* xn, uint32_t epoch, uint64_t timeout_ticks,
                                     uint32_t* status, int channel) {
    if (n < 1 || n > P2P_MAX_RANKS || hidden % 8 || hidden > P2P_NORM_MAX_HIDDEN || data_off % 16 || scratch_off % 16 || channel < 0 || channel >= P2P_CHANNELS)
        return hipErrorInvalidValue;
    if (rows == 0) return hipSuccess;
    const size_t flags_off = (size_t)channel * P2P_CHANNEL_FLAG_BYTES;
    const int64_t per = (rows + n - 1) / n;
    static const int cap = tune_int("PPLHIP_P2P_BLOCKS", 32);
    const int c = cap < 1 ? 1 : (cap > P2P_MAX_BLOCKS ? P2P_MAX_BLOCKS : cap);
    // one block per owned row up to the cap (the same count on every rank: the cross-rank barriers pair block b with block b); solo: every row
    const dim3 grid((unsigned)(n == 1 ? (rows < 256 ? rows : 256) : (per < c ? per : c))), block(512);
#define ARN(NN) hipLaunchKernelGGL(p2p_allreduce_norm_kernel<NN>, grid, block, 0, s, peers, me, n, data_off, scratch_off, rows, hidden / 8, hidden, \
                                   (uint4*)h, (const uint4*)w, eps, (uint4*)xn, epoch, (u64)timeout_ticks, status, flags_off)
    if (n == 1) ARN(1); else if (n == 2) ARN(2); else if (n == 4) ARN(4); else if (n == 8) ARN(8); else ARN(0);
#undef ARN
    return hipGetLastError();
}

hipError_t launch_p2p_allgather(hipStream_t s, const P2pPeers& peers, int me, int n, size_t src_off, void* dst, int64_t rows,
                                int64_t row_bytes, int64_t dst_row_bytes, uint32_t epoch, uint64_t timeout_ticks, uint32_t* status) {
    if (n < 2 || n > P2P_MAX_RANKS || row_bytes % 8 || dst_row_bytes % 8 || src_off % 8) return hipErrorInvalidValue;
    if (rows == 0 || row_bytes == 0) return hipSuccess;
    const bool wide = row_bytes % 16 == 0 && dst_row_bytes % 16 == 0 && src_off % 16 == 0 && (uintptr_t)dst % 16 == 0;
    const int gb = wide ? 16 : 8;
    const dim3 grid(grid_for(rows * row_bytes / gb)), block(512);
    if (wide)
        hipLaunchKernelGGL(p2p_allgather_kernel<16>, grid, block, 0, s, peers, me, n, src_off, reinterpret_cast<char*>(dst), rows, row_bytes / 16,
                           dst_row_bytes / 16, epoch, (u64)timeout_ticks, status);
    else
        hipLaunchKernelGGL(p2p_allgather_kernel<8>, grid, block, 0, s, peers, me, n, src_off, reinterpret_cast<char*>(dst), rows, row_bytes / 8,
                           dst_row_bytes / 8, epoch, (u64)timeout_ticks, status);
    return hipGetLastError();
}

}  // namespace pplhip
