// W8A16 linears of mid-size decode steps (4 < M <= 256) on FRAGMENT-MAJOR weights (round 4).
//
// These steps are bound by the weight stream, and what held the LDS-ring kernels of k_gemm.hip at 2.2-3.1 TB/s is the weight bytes a CU can
// keep in flight: every byte has to own LDS until it is consumed (32-64 KiB of the 160 KiB ring after the activation stages), and each
// 1-KiB LDS-DMA piece costs a wave ~60 issue cycles.  Here the weights never touch LDS.  A second copy of the matrix is laid out at load
// time in the order the matrix unit wants its operand (launch_pack_w8_frag): 1-KiB chunks of 32 output channels x 32 k, lane l of a wave
// owning the 16 bytes {W[n0 + l % 32][k0 + 8 (l / 32) + 0..7], W[n0 + l % 32][k0 + 16 + 8 (l / 32) + 0..7]} -- the B operands of the two
// v_mfma_f32_32x32x16_f16 of that chunk.  One global_load_dwordx4 per lane then fetches a whole contiguous KiB straight into the
// registers the conversion reads, the chunks of a 32-channel block are contiguous along K (a wave streams K * 32 consecutive bytes), and the
// bytes in flight are bounded by the register file (512 KiB per CU), not by LDS.  Only the activation tile (BM rows x 128 k, shared by
// the block's four waves) goes through LDS.
//
// Block: 4 waves, wave w multiplies NBW 32-channel blocks by the BM activation rows over the block's K slab; grid = channel blocks x
// row blocks x K slabs (fp32 split-K slabs, reduced by splitk_reduce_kernel or by the consuming kernel: kernels.h SplitSlabs).
// A operand = activations (rows m), B operand = weights: a lane's accumulators are 16 rows m of ONE channel n = n0 + l % 32, so a
// store instruction writes 32 consecutive channels of a row.
#include <algorithm>
#include "k_gemm_dev.h"

namespace pplhip {

typedef float f16v __attribute__((ext_vector_type(16)));

// ---- load-time repack: W [N][K] int8 row-major -> fragment-major, channels padded to a multiple of 32 with zeros ----------------------
__global__ __launch_bounds__(256) void pack_w8_frag_kernel(const int8_t* __restrict__ w, int N, int K, uint4* __restrict__ out) {
    const int64_t total = (int64_t)((N + 31) / 32) * (K / 32) * 64;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int l = (int)(i & 63);
        const int64_t chunk = i >> 6;
        const int kc = (int)(chunk % (K / 32));
        const int nb = (int)(chunk / (K / 32));
        const int n = nb * 32 + (l & 31);
        uint2 lo = make_uint2(0, 0), hi = make_uint2(0, 0);
        if (n < N) {
            const int8_t* p = w + (int64_t)n * K + kc * 32 + (l >> 5) * 8;
            lo = *reinterpret_cast<const uint2*>(p);
            hi = *reinterpret_cast<const uint2*>(p + 16);
        }
        out[i] = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
}

size_t w8_frag_bytes(int N, int K) { return (size_t)((N + 31) / 32) * 32 * (size_t)K; }

hipError_t launch_pack_w8_frag(hipStream_t s, const int8_t* w, int N, int K, void* out) {
    if (K % 32 || ((uintptr_t)w & 7) || K % 8) return hipErrorInvalidValue;
    const int64_t total = (int64_t)((N + 31) / 32) * (K / 32) * 64;
    if (total == 0) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 65536);
    hipLaunchKernelGGL(pack_w8_frag_kernel, dim3(blocks), dim3(256), 0, s, w, N, K, (uint4*)out);
    return hipGetLastError();
}

// ---- the GEMM ------------------------------------------------------------------------------------------------------------------------
constexpr int F_KT = 128;                 // k per activation tile (4 weight chunks)
constexpr int F_ROWB = F_KT * 2;          // LDS bytes of an activation row (16 chunks of 16 B, XOR-swizzled with row % 16)
template <int BM> constexpr int f_xd() { return BM <= 64 ? 2 : 1; }                     // activation tiles in flight ahead of the one being multiplied
template <int BM> constexpr int f_lds() { return (f_xd<BM>() + 1) * BM * F_ROWB; }     // 24 / 48 / 64 KiB

template <int I> struct IC { static constexpr int value = I; };

// one 16-byte weight load per lane, issued from inline asm (the kernel counts vmcnt by hand); OFF: immediate byte offset
template <int OFF>
__device__ __forceinline__ void frag_wload(kv_u32x4& dst, uint32_t voff, const char* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 nt" : "=&v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
// wait until at most N vector-memory operations are in flight; the registers are operands so that no use of them is scheduled in front
template <int N>
__device__ __forceinline__ void frag_wwait(kv_u32x4& a, kv_u32x4& b) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void frag_wwait(kv_u32x4& a) {
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void frag_vmwait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int EPI, int BM, int NBW>
__global__ __launch_bounds__(256, BM * NBW <= 128 ? 2 : 1) void gemm_w8_frag_kernel(
    const uint16_t* __restrict__ x, const uint4* __restrict__ wf, const uint16_t* __restrict__ scale, int64_t M, int N, int K,
    void* __restrict__ yv, int64_t ldy, int nblocks32 /* ceil(N / 32) */, int m_blocks, int kt_per_split, float* __restrict__ ws, int dbg) {
    constexpr int NJ = BM / 32, XPW = BM / 16, XB = BM * F_ROWB, XD = f_xd<BM>(), XST = XD + 1;
    extern __shared__ __attribute__((aligned(16))) char smem_f[];   // XST activation stages (ONE shared object: k_gemm_dev.h)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // block id -> (channel block group cb, row block mb): the row blocks of one channel group sit 8 ids apart = on the same XCD, whose L2
    // then serves the second reader of the weights
    const int id = blockIdx.x;
    const int mb = (id >> 3) % m_blocks;
    const int cb = (id / (8 * m_blocks)) * 8 + (id & 7);
    if (cb * 4 * NBW >= nblocks32) return;                 // (the grid is padded to whole rounds of the 8 XCDs)
    const int nb0 = (cb * 4 + wave) * NBW;                 // first 32-channel block of this wave
    const int64_t m0 = (int64_t)mb * BM;
    const int split_id = blockIdx.y, n_splits = gridDim.y;
    const int kt_all = K / F_KT;
    const int kt0 = split_id * kt_per_split;
    const int ktiles = (kt0 + kt_per_split < kt_all) ? kt_per_split : kt_all - kt0;
    const int kchunks = K / 32;

    // activation tile by LDS-DMA: piece P = wave + 4 q is rows 4 P .. 4 P + 3 (256 B each); the lane that lands at 16-byte position pos of
    // row r fetches chunk pos ^ (r % 16), and the fragment reads below undo the permutation (no bank conflicts: 16 rows -> 16 positions)
    const uint32_t xbase = lds_addr(smem_f);
    const char* xsrc[XPW];
    uint32_t xdst[XPW];
#pragma unroll
    for (int q = 0; q < XPW; ++q) {
        const int P = wave + 4 * q, row = 4 * P + (lane >> 4), chunk = (lane & 15) ^ (row & 15);
        int64_t m = m0 + row;
        if (m >= M) m = M - 1;
        xsrc[q] = reinterpret_cast<const char*>(x + m * K + (int64_t)kt0 * F_KT + chunk * 8);
        xdst[q] = __builtin_amdgcn_readfirstlane(xbase + P * 1024);
    }
    auto xissue = [&](int t, int stage) {   // tile t of this block -> stage t % XST
        const uint32_t so = (uint32_t)stage * XB;
#pragma unroll
        for (int q = 0; q < XPW; ++q) glds16(xsrc[q] + (int64_t)t * (F_KT * 2), xdst[q] + so);
    };
    // weight stream of this wave: NBW channel blocks, 4 chunks per tile, three register sets (two tiles in flight).  The loads are
    // issued from inline asm and waited for by hand (wwait): hipcc cannot see the activation DMA, so its own counts would wait for
    // every operation older than the newest DMA -- a whole tile of prefetch distance lost
    const char* wsrc[NBW];
    bool wlive[NBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
        wlive[i] = nb0 + i < nblocks32;   // (wave-uniform)
        const int nb = wlive[i] ? nb0 + i : nblocks32 - 1;
        wsrc[i] = reinterpret_cast<const char*>(wf) + ((int64_t)nb * kchunks + (int64_t)kt0 * 4) * 1024;
    }
    const uint32_t wlane = lane * 16;
    kv_u32x4 wr[3][4][NBW];
    auto wissue = [&](auto S_, int t) {
        constexpr int S = decltype(S_)::value;
        const char* base[NBW];
#pragma unroll
        for (int i = 0; i < NBW; ++i) base[i] = wsrc[i] + (int64_t)t * 4096;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < NBW; ++i)
                if (c == 0) frag_wload<0>(wr[S][c][i], wlane, base[i]);
                else if (c == 1) frag_wload<1024>(wr[S][c][i], wlane, base[i]);
                else if (c == 2) frag_wload<2048>(wr[S][c][i], wlane, base[i]);
                else frag_wload<3072>(wr[S][c][i], wlane, base[i]);
    };
    // fragment read offsets: row lane % 32 (+ 32 j), chunk 4 c + 2 h + lane / 32, swizzled
    uint32_t xoff[8];
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) xoff[ch] = (lane & 31) * F_ROWB + ((((lane & 15) ^ (lane >> 5)) ^ (2 * ch)) << 4);

    f16v acc[NBW][NJ];
#pragma unroll
    for (int i = 0; i < NBW; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // IW / IX: this step issues the weights of tile t + 2 / the activations of tile t + XD; PW / PX: what the step before it issued.
    // Compile-time flags, so that every wait is an exact count of the vector-memory operations younger than the one it needs
    // (issue order of a step: activation DMA pieces, then the weight loads chunk by chunk)
    auto step = [&](auto S_, auto IW_, auto IX_, auto PW_, auto PX_, int t) {
        constexpr int S = decltype(S_)::value;
        constexpr bool IW = decltype(IW_)::value != 0, IX = decltype(IX_)::value != 0, PW = decltype(PW_)::value != 0, PX = decltype(PX_)::value != 0;
        constexpr int CUR = (IW ? 4 * NBW : 0) + (IX ? XPW : 0);
        constexpr int PREV = (PW ? 4 * NBW : 0) + (PX && XD == 2 ? XPW : 0);   // (XD == 1: the previous step's DMA was waited for at its end)
        if (dbg & 2) __syncthreads();
        if constexpr (IX) xissue(t + XD, XST == 3 ? (S + 2) % 3 : ((t + 1) & 1));
        if constexpr (IW) wissue(IC<(S + 2) % 3>{}, t + 2);
        const char* xs = smem_f + (XST == 3 ? S : (t & 1)) * XB;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            // chunk c of tile t: its later chunks and everything issued since may stay in flight.  The registers are operands of the
            // wait so that no use of them can be scheduled in front of it
            if (dbg & 1) frag_vmwait<0>();
#define FRAG_WAIT(C) do { if constexpr (NBW == 1) frag_wwait<(3 - C) * NBW + PREV + CUR>(wr[S][C][0]); else frag_wwait<(3 - C) * NBW + PREV + CUR>(wr[S][C][0], wr[S][C][NBW - 1]); } while (0)
            if (c == 0) FRAG_WAIT(0); else if (c == 1) FRAG_WAIT(1); else if (c == 2) FRAG_WAIT(2); else FRAG_WAIT(3);
#undef FRAG_WAIT
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                h8 a[NJ], b[NBW];
#pragma unroll
                for (int j = 0; j < NJ; ++j) a[j] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(xs + j * 32 * F_ROWB + xoff[c * 2 + h]));
#pragma unroll
                for (int i = 0; i < NBW; ++i) b[i] = cvt_i8x8_f16(h == 0 ? make_uint2(wr[S][c][i][0], wr[S][c][i][1]) : make_uint2(wr[S][c][i][2], wr[S][c][i][3]));
#pragma unroll
                for (int i = 0; i < NBW; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j], b[i], acc[i][j], 0, 0, 0);
            }
        }
        // the activations of tile t + 1 must have landed before the barrier publishes them
        constexpr int YOUNGER = XD == 2 ? (PW ? 4 * NBW : 0) + CUR : (IW ? 4 * NBW : 0);
        if (dbg & 4) frag_vmwait<0>();
        frag_vmwait<YOUNGER>();
        __syncthreads();
    };
    constexpr IC<1> Y{};
    constexpr IC<0> N_{};
    constexpr IC<XD == 2 ? 0 : 1> X1{};   // "activations only": the step before the last one when XD == 1

    // the last steps of a slab: rem = 2 .. 4 tiles left, tiles t and t + 1 in flight
    auto tail = [&](int t, int rem) {
        if (rem == 4) {
            step(IC<0>{}, Y, Y, Y, Y, t);
            step(IC<1>{}, Y, Y, Y, Y, t + 1);
            step(IC<2>{}, N_, X1, Y, Y, t + 2);
            step(IC<0>{}, N_, N_, N_, X1, t + 3);
        } else if (rem == 3) {
            step(IC<0>{}, Y, Y, Y, Y, t);
            step(IC<1>{}, N_, X1, Y, Y, t + 1);
            step(IC<2>{}, N_, N_, N_, X1, t + 2);
        } else {
            step(IC<0>{}, N_, X1, Y, Y, t);
            step(IC<1>{}, N_, N_, N_, X1, t + 1);
        }
    };
    if (ktiles >= 2) {
        xissue(0, 0);
        wissue(IC<0>{}, 0);
        if (XD == 2) xissue(1, 1);
        wissue(IC<1>{}, 1);
        frag_vmwait<4 * NBW + (XD == 2 ? XPW : 0)>();
        __syncthreads();
        int t = 0;
        for (; t + 5 <= ktiles; t += 3) {   // all three steps issue: t + 2 + 2 < ktiles
            step(IC<0>{}, Y, Y, Y, Y, t);
            step(IC<1>{}, Y, Y, Y, Y, t + 1);
            step(IC<2>{}, Y, Y, Y, Y, t + 2);
        }
        tail(t, ktiles - t);
    } else if (ktiles == 1) {
        xissue(0, 0);
        wissue(IC<0>{}, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        step(IC<0>{}, N_, N_, N_, N_, 0);
    }

    // accumulator r of tile j: row m0 + 32 j + 8 (r / 4) + 4 (lane / 32) + r % 4, channel 32 nb + lane % 32
    if (n_splits > 1) {
        float* slab = ws + (int64_t)split_id * M * N;
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
            const int n = (nb0 + i) * 32 + (lane & 31);
            if (!wlive[i] || n >= N) continue;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t m = m0 + j * 32 + (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3);
                    if (m < M) slab[m * N + n] = acc[i][j][r];
                }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
        const int n = (nb0 + i) * 32 + (lane & 31);
        const bool live = wlive[i] && n < N;
        const float sc = live ? h2f(scale[n]) : 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + j * 32 + (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3);
                const float v = acc[i][j][r] * sc;
                if constexpr (EPI == EPI_SWIGLU) {
                    // channels (gate, up) are neighbours: the even lane takes the odd lane's value
                    const float g = round_h(v), u = round_h(__shfl_xor(v, 1, 64));
                    if (live && m < M && !(lane & 1)) reinterpret_cast<uint16_t*>(yv)[m * ldy + (n >> 1)] = f2h(g / (1.0f + __expf(-g)) * u);
                } else if constexpr (EPI == EPI_F32) {
                    if (live && m < M) reinterpret_cast<float*>(yv)[m * ldy + n] = v;
                } else {
                    if (live && m < M) reinterpret_cast<uint16_t*>(yv)[m * ldy + n] = f2h(v);
                }
            }
    }
}

// shapes the kernel takes
bool linear_w8_frag_supported(int64_t M, int N, int K) { return M >= 1 && M <= 256 && K % F_KT == 0 && N % 4 == 0; }

hipError_t launch_linear_w8_frag(hipStream_t s, const uint16_t* x, const void* wfrag, const uint16_t* scale, int64_t M, int N, int K, void* y,
                                 int64_t ldy, int epi, float* ws, size_t ws_bytes, SplitSlabs* defer) {
    if (defer) *defer = SplitSlabs{};
    if (!linear_w8_frag_supported(M, N, K)) return hipErrorInvalidValue;
    static const int f_bm = getenv("PPLHIP_FRAG_BM") ? atoi(getenv("PPLHIP_FRAG_BM")) : 0;
    static const int f_nbw = getenv("PPLHIP_FRAG_NBW") ? atoi(getenv("PPLHIP_FRAG_NBW")) : 0;
    static const int f_split = getenv("PPLHIP_FRAG_SPLITK") ? atoi(getenv("PPLHIP_FRAG_SPLITK")) : 0;
    static const int f_dbg = getenv("PPLHIP_FRAG_DBG") ? atoi(getenv("PPLHIP_FRAG_DBG")) : 4;  // 4: drain vmcnt before every publishing barrier (the exact count under-waits: wrong results, profiles/r04_frag_experiment.md)
    static const int f_target = getenv("PPLHIP_FRAG_BLOCKS") ? atoi(getenv("PPLHIP_FRAG_BLOCKS")) : 512;
    int bm = M <= 32 ? 32 : (M <= 64 ? 64 : 128);
    if (f_bm == 32 || f_bm == 64 || f_bm == 128) bm = f_bm;
    // (NBW = 2 -- two channel blocks per wave, each activation fragment feeding two products -- needs more than 256 registers with three
    // weight sets; hipcc then spills registers that are targets of loads in flight.  Not instantiated.)
    const int nbw = 1;
    (void)f_nbw;
    const int nblocks32 = (N + 31) / 32;
    const int cbs = (nblocks32 + 4 * nbw - 1) / (4 * nbw);
    const int cbs_pad = (cbs + 7) / 8 * 8;
    const int m_blocks = (int)((M + bm - 1) / bm);
    const int kt_all = K / F_KT;
    int splits = 1;
    if (ws && ws_bytes) {
        splits = (f_target + cbs * m_blocks - 1) / (cbs * m_blocks);
        if (splits > 8) splits = 8;
        if (splits > kt_all / 2) splits = kt_all / 2 > 0 ? kt_all / 2 : 1;
        if (f_split > 0) splits = f_split;
        while (splits > 1 && (size_t)splits * M * N * sizeof(float) > ws_bytes) --splits;
    }
    const int kt_per = (kt_all + splits - 1) / splits;
    splits = (kt_all + kt_per - 1) / kt_per;
    dim3 g((unsigned)(cbs_pad * m_blocks), (unsigned)splits);
#define FR_L(E, B, W) hipLaunchKernelGGL((gemm_w8_frag_kernel<E, B, W>), g, dim3(256), f_lds<B>(), s, x, (const uint4*)wfrag, scale, M, N, K, y, ldy, nblocks32, m_blocks, kt_per, ws, f_dbg)
#define FR_B(E) do { if (bm == 32) FR_L(E, 32, 1); else if (bm == 64) FR_L(E, 64, 1); else FR_L(E, 128, 1); } while (0)
    if (epi == EPI_F32) FR_B(EPI_F32); else if (epi == EPI_F16) FR_B(EPI_F16); else FR_B(EPI_SWIGLU);
#undef FR_B
#undef FR_L
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || splits == 1) return e;
    if (defer && epi == EPI_F16 && N % 8 == 0 && splits <= 8) { *defer = SplitSlabs{ws, splits, scale, N, M}; return e; }
    return launch_splitk_reduce(s, ws, splits, M, N, scale, y, ldy, epi);
}

}  // namespace pplhip
