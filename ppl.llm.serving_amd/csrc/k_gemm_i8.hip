// online_i8i8 (W8A8) linear layers: the quant method src/backends/cuda/resource_manager.cc:51-52 hands to ppl.nn.
//   activations  per token row:   sx = max|x| / 127 (fp32), q = clamp(rint(x * (127 / max|x|)))           quant_act_kernel
//   weights      per output row:  scale = fp16(max|w| / 127), q = clamp(rint(w / scale))   (once, at load) quant_weight_kernel
//   y[m,n] = fp16( (float)(sum_k qx * qw) * sx[m] * scale[n] ),  the sum exact in int32 on the matrix cores
// Oracle: linear_fwd_a8 / ref_quant_act_rows / ref_quant_weight_rows (oracle/llama_ref.c) -- integer accumulation makes
// the GEMM itself bit-exact against it; only the fp16 rounding of the two fp32 multiplies is left, done in the same order.
//
// Tile kernel (M > 32, K % 128 == 0): block tile 128 (n) x 128 (m) x 128 (k) int8, 4 waves as 2 x 2, each 64 x 64 =
// 4 x 4 tiles of v_mfma_i32_16x16x64_i8 (weights = A operand, so a lane owns 4 consecutive n of one activation row).
// Both operands go global -> LDS by DMA into [row][128 B] tiles whose 16-byte chunk index is XOR-swizzled with
// (row >> 1) & 7 on the source address and on the fragment reads (same scheme as the fp16 activation tile of k_gemm_dev.h).
// Two LDS stages (64 KiB -> two blocks per CU), one barrier per K tile; launches of <= 256 tiles use the 8-wave producer /
// consumer form (gemm_i8_pc_kernel).
// Skinny / generic kernel (any M, K % 16 == 0): one block per 16 weight rows, waves split K, weights straight from HBM
// into the MFMA A operand, activations (L2 resident) into B; grid.y walks 32-row activation groups.
#include <stdlib.h>

#include <algorithm>

#include "k_gemm_dev.h"

namespace pplhip {

typedef int i4v __attribute__((ext_vector_type(4)));

constexpr int I_BN = 128, I_BM = 128, I_BK = 128;

__device__ __forceinline__ float block_max_256(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__device__ __forceinline__ int q8(float v) {
    v = rintf(v);
    v = fminf(fmaxf(v, -127.f), 127.f);
    return (int)v;
}

// one block per row; x [M, ldx] fp16 (K valid), q [M, ldq] int8 (columns K..ldq-1 zero), sx [M]
__global__ __launch_bounds__(256) void quant_act_kernel(const uint16_t* __restrict__ x, int K, int64_t ldx, int8_t* __restrict__ q,
                                                        int64_t ldq, float* __restrict__ sx) {
    __shared__ float red[4];
    const int64_t m = blockIdx.x;
    const uint16_t* xr = x + m * ldx;
    int8_t* qr = q + m * ldq;
    const int K8 = K >> 3;
    float amax = 0.f;
    for (int i = threadIdx.x; i < K8; i += 256) {
        const h8 v = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(xr + i * 8));
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf((float)v[j]));
    }
    for (int k = K8 * 8 + threadIdx.x; k < K; k += 256) amax = fmaxf(amax, fabsf(h2f(xr[k])));
    amax = block_max_256(amax, red);
    const float inv = amax > 0.f ? 127.0f / amax : 0.f;
    if (threadIdx.x == 0) sx[m] = amax / 127.0f;
    for (int i = threadIdx.x; i < K8; i += 256) {
        const h8 v = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(xr + i * 8));
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) lo |= (uint32_t)(q8((float)v[j] * inv) & 0xff) << (8 * j);
#pragma unroll
        for (int j = 0; j < 4; ++j) hi |= (uint32_t)(q8((float)v[4 + j] * inv) & 0xff) << (8 * j);
        *reinterpret_cast<uint2*>(qr + i * 8) = make_uint2(lo, hi);
    }
    for (int k = K8 * 8 + threadIdx.x; k < K; k += 256) qr[k] = (int8_t)q8(h2f(xr[k]) * inv);
    for (int64_t k = K + threadIdx.x; k < ldq; k += 256) qr[k] = 0;
}

// one block per weight row; w [N, K] fp16 -> q [N, ldq] int8 (pad columns zero) + scale [N] fp16
__global__ __launch_bounds__(256) void quant_weight_kernel(const uint16_t* __restrict__ w, int K, int8_t* __restrict__ q, int64_t ldq,
                                                           uint16_t* __restrict__ scale) {
    __shared__ float red[4];
    const int64_t n = blockIdx.x;
    const uint16_t* wr = w + n * K;
    int8_t* qr = q + n * ldq;
    float amax = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) amax = fmaxf(amax, fabsf(h2f(wr[k])));
    amax = block_max_256(amax, red);
    const uint16_t sh = f2h(amax / 127.0f);
    if (threadIdx.x == 0) scale[n] = sh;
    const float s = h2f(sh) > 0.f ? h2f(sh) : 1.0f;
    for (int k = threadIdx.x; k < K; k += 256) qr[k] = (int8_t)q8(h2f(wr[k]) / s);  // IEEE division (hipcc default: correctly rounded)
    for (int64_t k = K + threadIdx.x; k < ldq; k += 256) qr[k] = 0;
}

template <int EPI>
__device__ __forceinline__ void store4_i8(void* yv, int64_t ldy, int64_t m, int n, i4v acc, float sxm, h4 sh) {
    store4<EPI>(yv, ldy, m, n, ((float)acc[0] * sxm) * (float)sh[0], ((float)acc[1] * sxm) * (float)sh[1],
                ((float)acc[2] * sxm) * (float)sh[2], ((float)acc[3] * sxm) * (float)sh[3]);
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_i8_kernel(const int8_t* __restrict__ xq, const float* __restrict__ sx,
                                                         const int8_t* __restrict__ w, const uint16_t* __restrict__ scale, int64_t M,
                                                         int N, int K, void* __restrict__ yv, int64_t ldy, int n_tiles, int m_tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem_i8[];  // 2 x (X 16 KiB + W 16 KiB)
    constexpr int TILE = I_BM * I_BK;                               // bytes of one operand tile
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;  // the M tiles of one weight tile run on one XCD (block b -> XCD b % 8)
    const int nt = xcd + 8 * (slot / m_tiles);
    const int mt = slot % m_tiles;
    if (nt >= n_tiles) return;
    const int n0 = nt * I_BN;
    const int64_t m0 = (int64_t)mt * I_BM;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int wn = wave & 1, wm = wave >> 1;

    const int8_t* xsrc[4];
    const int8_t* wsrc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = j * 256 + tid, row = p >> 3, c = (p & 7) ^ ((row >> 1) & 7);
        int64_t m = m0 + row;
        if (m >= M) m = M - 1;
        int n = n0 + row;
        if (n >= N) n = N - 1;
        xsrc[j] = xq + m * K + c * 16;
        wsrc[j] = w + (int64_t)n * K + c * 16;
    }
    const uint32_t xdst = __builtin_amdgcn_readfirstlane(lds_addr(smem_i8) + wave * 1024);
    const uint32_t wdst = xdst + 2 * TILE;
    auto issue = [&](int stage, int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(xsrc[j] + k0, xdst + stage * TILE + j * 4096);
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(wsrc[j] + k0, wdst + stage * TILE + j * 4096);
    };

    i4v acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = i4v{0, 0, 0, 0};

    const int ktiles = K / I_BK;
    issue(0, 0);
    for (int t = 0; t < ktiles; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < ktiles) issue((t + 1) & 1, (t + 1) * I_BK);
        const char* xs = smem_i8 + (t & 1) * TILE;
        const char* ws = smem_i8 + 2 * TILE + (t & 1) * TILE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            i4v a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wn * 64 + i * 16 + l15;
                a[i] = *reinterpret_cast<const i4v*>(ws + row * I_BK + g_swz(row, ks * 4 + kq) * 16);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wm * 64 + j * 16 + l15;
                b[j] = *reinterpret_cast<const i4v*>(xs + row * I_BK + g_swz(row, ks * 4 + kq) * 16);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }

#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t m = m0 + wm * 64 + j * 16 + l15;
        if (m >= M) continue;
        const float sxm = sx[m];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + i * 16 + kq * 4;
            if (n >= N) continue;
            const h4 sh = __builtin_bit_cast(h4, *reinterpret_cast<const uint2*>(scale + n));
            store4_i8<EPI>(yv, ldy, m, n, acc[i][j], sxm, sh);
        }
    }
}

// Producer / consumer form of the tile kernel: 8 waves, waves 0..3 multiply (2 x 2 layout as above), waves 4..7 do nothing but wait
// for their LDS-DMA pieces and refill the ring (the ~100-cycle issue cost of a piece then runs beside the MFMA stream instead of in
// front of it: int8 moves 8 pieces per 32 MFMAs and wave, more than the fp16 kernel).  ST stages of 32 KiB, one barrier per tile.
template <int EPI, int ST>
__global__ __launch_bounds__(512) void gemm_i8_pc_kernel(const int8_t* __restrict__ xq, const float* __restrict__ sx,
                                                         const int8_t* __restrict__ w, const uint16_t* __restrict__ scale, int64_t M,
                                                         int N, int K, void* __restrict__ yv, int64_t ldy, int n_tiles, int m_tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem_i8[];  // ST x (X 16 KiB) then ST x (W 16 KiB)
    constexpr int TILE = I_BM * I_BK;
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int nt = xcd + 8 * (slot / m_tiles);
    const int mt = slot % m_tiles;
    if (nt >= n_tiles) return;
    const int n0 = nt * I_BN;
    const int64_t m0 = (int64_t)mt * I_BM;
    const bool producer = threadIdx.x >= 256;
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int ktiles = K / I_BK;
    constexpr int D = ST - 1, PT = 8;  // pieces per producer wave and tile

    if (producer) {
        const int8_t* xsrc[4];
        const int8_t* wsrc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = j * 256 + tid, row = p >> 3, c = (p & 7) ^ ((row >> 1) & 7);
            int64_t m = m0 + row;
            if (m >= M) m = M - 1;
            int n = n0 + row;
            if (n >= N) n = N - 1;
            xsrc[j] = xq + m * K + c * 16;
            wsrc[j] = w + (int64_t)n * K + c * 16;
        }
        const uint32_t xdst = __builtin_amdgcn_readfirstlane(lds_addr(smem_i8) + wave * 1024);
        const uint32_t wdst = xdst + ST * TILE;
        auto issue = [&](int stage, int k0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) glds16(xsrc[j] + k0, xdst + stage * TILE + j * 4096);
#pragma unroll
            for (int j = 0; j < 4; ++j) glds16(wsrc[j] + k0, wdst + stage * TILE + j * 4096);
        };
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (d < ktiles) issue(d, d * I_BK);
        int stn = D % ST;
        for (int t = 0; t < ktiles; ++t) {
            const int younger = (ktiles - 1 - t) < (D - 1) ? (ktiles - 1 - t) : (D - 1);
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PT) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t + D < ktiles) issue(stn, (t + D) * I_BK);
            stn = stn == ST - 1 ? 0 : stn + 1;
        }
        return;
    }

    const int wn = wave & 1, wm = wave >> 1;
    i4v acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = i4v{0, 0, 0, 0};
    int st = 0;
    for (int t = 0; t < ktiles; ++t) {
        __syncthreads();
        const char* xs = smem_i8 + st * TILE;
        const char* ws = smem_i8 + ST * TILE + st * TILE;
        st = st == ST - 1 ? 0 : st + 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            i4v a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wn * 64 + i * 16 + l15;
                a[i] = *reinterpret_cast<const i4v*>(ws + row * I_BK + g_swz(row, ks * 4 + kq) * 16);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wm * 64 + j * 16 + l15;
                b[j] = *reinterpret_cast<const i4v*>(xs + row * I_BK + g_swz(row, ks * 4 + kq) * 16);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t m = m0 + wm * 64 + j * 16 + l15;
        if (m >= M) continue;
        const float sxm = sx[m];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + i * 16 + kq * 4;
            if (n >= N) continue;
            const h4 sh = __builtin_bit_cast(h4, *reinterpret_cast<const uint2*>(scale + n));
            store4_i8<EPI>(yv, ldy, m, n, acc[i][j], sxm, sh);
        }
    }
}

// 128 (m) x 384 (n) x 128 (k) block: the int8 form of gemm_w8_wide_kernel (k_gemm_wide.hip).  Twelve consumer waves as 6 (n) x 2 (m), wave
// tile 64 x 64 like the waves of the kernel above, share ONE activation tile; four producer waves keep a two-stage ring of 64 KiB filled.
// Two co-resident 128 x 128 blocks move 64 KiB per 1024 MFMA cycles and CU -- the whole 64 B / clk of the CU's vector-memory path --, this
// block moves 64 KiB per 3072 (profiles/r03_gemm_experiments.md #16-17).  Taken when its tiles fill rounds of 256 one-per-CU blocks.
constexpr int IW_BN = 384, IW_NC = 12, IW_NP = 4, IW_ST = 2;
constexpr int IW_XB = I_BM * I_BK, IW_WB = IW_BN * I_BK;       // 16 KiB + 48 KiB per stage
constexpr int IW_PP = (IW_XB + IW_WB) / 1024 / IW_NP;           // 16 one-KiB pieces per producer wave and tile (4 activation + 12 weight)
template <int EPI>
__global__ __launch_bounds__((IW_NC + IW_NP) * 64) void gemm_i8_wide_kernel(const int8_t* __restrict__ xq, const float* __restrict__ sx,
                                                                             const int8_t* __restrict__ w, const uint16_t* __restrict__ scale,
                                                                             int64_t M, int N, int K, void* __restrict__ yv, int64_t ldy,
                                                                             int n_tiles, int m_tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem_i8[];  // IW_ST x (X 16 KiB) then IW_ST x (W 48 KiB)
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int nt = xcd + 8 * (slot / m_tiles);
    const int mt = slot % m_tiles;
    if (nt >= n_tiles) return;
    const int n0 = nt * IW_BN;
    const int64_t m0 = (int64_t)mt * I_BM;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ktiles = K / I_BK;

    if (wave >= IW_NC) {
        // producer pw: pieces P = pw + 4 j; j < 4: activation piece P (rows 8 P .. + 8), else weight piece P - 16 (rows 8 (P - 16) .. + 8);
        // 128-byte rows, 16-byte chunks XOR-swizzled with (row >> 1) & 7 on the SOURCE side (the DMA writes LDS linearly)
        const int pw = wave - IW_NC;
        const int8_t* psrc[IW_PP];
        uint32_t pdst[IW_PP];
        const uint32_t xbase = lds_addr(smem_i8), wbase = xbase + IW_ST * IW_XB;
#pragma unroll
        for (int j = 0; j < IW_PP; ++j) {
            const int P = pw + IW_NP * j;
            const bool isx = j < IW_XB / 1024 / IW_NP;
            const int Pl = isx ? P : P - IW_XB / 1024;
            const int p = Pl * 64 + lane, row = p >> 3, c = (p & 7) ^ ((row >> 1) & 7);
            if (isx) {
                int64_t m = m0 + row;
                if (m >= M) m = M - 1;
                psrc[j] = xq + m * K + c * 16;
                pdst[j] = __builtin_amdgcn_readfirstlane(xbase + Pl * 1024);
            } else {
                int n = n0 + row;
                if (n >= N) n = N - 1;
                psrc[j] = w + (int64_t)n * K + c * 16;
                pdst[j] = __builtin_amdgcn_readfirstlane(wbase + Pl * 1024);
            }
        }
#define IW_PRODUCE(KT, STG)                                                                                                  \
    do {                                                                                                                     \
        _Pragma("unroll") for (int j = 0; j < IW_PP; ++j)                                                                    \
            glds16(psrc[j] + (int64_t)(KT) * I_BK, pdst[j] + (STG) * (j < IW_XB / 1024 / IW_NP ? IW_XB : IW_WB));            \
    } while (0)
        IW_PRODUCE(0, 0);
        for (int t = 0; t < ktiles; ++t) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();  // tile t is published; the other stage (read during iteration t - 1) is free
            if (t + 1 < ktiles) IW_PRODUCE(t + 1, (t + 1) & 1);
        }
#undef IW_PRODUCE
        return;
    }

    const int l15 = lane & 15, kq = lane >> 4;
    const int wn = wave % 6, wm = wave / 6;
    i4v acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = i4v{0, 0, 0, 0};
    for (int t = 0; t < ktiles; ++t) {
        __syncthreads();
        const char* xs = smem_i8 + (t & 1) * IW_XB;
        const char* ws = smem_i8 + IW_ST * IW_XB + (t & 1) * IW_WB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            i4v a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wn * 64 + i * 16 + l15;
                a[i] = *reinterpret_cast<const i4v*>(ws + row * I_BK + g_swz(row, ks * 4 + kq) * 16);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wm * 64 + j * 16 + l15;
                b[j] = *reinterpret_cast<const i4v*>(xs + row * I_BK + g_swz(row, ks * 4 + kq) * 16);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t m = m0 + wm * 64 + j * 16 + l15;
        if (m >= M) continue;
        const float sxm = sx[m];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + i * 16 + kq * 4;
            if (n >= N) continue;
            const h4 sh = __builtin_bit_cast(h4, *reinterpret_cast<const uint2*>(scale + n));
            store4_i8<EPI>(yv, ldy, m, n, acc[i][j], sxm, sh);
        }
    }
}

// Large M (steps that carry prefill): 256(n) x 256(m) x 64(k) tiles, 8 waves as 4 (n) x 2 (m), wave tile 64 x 128 = 4 x 8 MFMA tiles
// (12 fragment reads and 4 LDS-DMA pieces per 32 MFMAs and wave, against 16 and 8 in the 128 x 128 kernels), 64-byte rows with the
// w_swz chunk swizzle for both operands, ST-stage ring of 32 KiB, one barrier per tile, one block per CU.
template <int EPI, int ST>
__global__ __launch_bounds__(512) void gemm_i8_256_kernel(const int8_t* __restrict__ xq, const float* __restrict__ sx,
                                                          const int8_t* __restrict__ w, const uint16_t* __restrict__ scale, int64_t M,
                                                          int N, int K, void* __restrict__ yv, int64_t ldy, int n_tiles, int m_tiles,
                                                          int gn, int gm) {
    extern __shared__ __attribute__((aligned(16))) char smem_i8[];  // ST x (X 16 KiB) then ST x (W 16 KiB)
    constexpr int TILE = 256 * 64;
    // XCD id % 8 walks its weight tiles in super-tiles of gn (n) x gm (m) tiles (k_gemm.hip, gemm_w8_dma256_kernel): the blocks it runs at a
    // time share gn weight and gm activation tiles per K step
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int per_super = gn * gm, sm_count = m_tiles / gm;
    const int sup = slot / per_super, within = slot % per_super;
    const int nt = xcd + 8 * ((sup / sm_count) * gn + within / gm);
    const int mt = (sup % sm_count) * gm + within % gm;
    if (nt >= n_tiles) return;
    const int n0 = nt * 256;
    const int64_t m0 = (int64_t)mt * 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int wn = wave & 3, wm = wave >> 2;

    const int8_t* xsrc[2];
    const int8_t* wsrc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = j * 512 + tid, row = p >> 2, c = (p & 3) ^ w_swz(row);
        int64_t m = m0 + row;
        if (m >= M) m = M - 1;
        int n = n0 + row;
        if (n >= N) n = N - 1;
        xsrc[j] = xq + m * K + c * 16;
        wsrc[j] = w + (int64_t)n * K + c * 16;
    }
    const uint32_t xdst = __builtin_amdgcn_readfirstlane(lds_addr(smem_i8) + wave * 1024);
    const uint32_t wdst = xdst + ST * TILE;
    auto issue = [&](int stage, int k0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(xsrc[j] + k0, xdst + stage * TILE + j * 8192);
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(wsrc[j] + k0, wdst + stage * TILE + j * 8192);
    };
    int woff[4], xoff[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wn * 64 + i * 16 + l15;
        woff[i] = ST * TILE + row * 64 + (kq ^ w_swz(row)) * 16;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = wm * 128 + j * 16 + l15;
        xoff[j] = row * 64 + (kq ^ w_swz(row)) * 16;
    }
    i4v acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = i4v{0, 0, 0, 0};

    constexpr int D = ST - 1, PT = 4;
    const int ktiles = K / 64;
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < ktiles) issue(d, d * 64);
    int st = 0, stn = D % ST;
    for (int t = 0; t < ktiles; ++t) {
        const int younger = (ktiles - 1 - t) < (D - 1) ? (ktiles - 1 - t) : (D - 1);
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PT) : "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + D < ktiles) issue(stn, (t + D) * 64);
        const char* sb = smem_i8 + st * TILE;
        st = st == ST - 1 ? 0 : st + 1;
        stn = stn == ST - 1 ? 0 : stn + 1;
        i4v a[4], b[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const i4v*>(sb + woff[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) b[j] = *reinterpret_cast<const i4v*>(sb + xoff[j]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i], b[j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t m = m0 + wm * 128 + j * 16 + l15;
        if (m >= M) continue;
        const float sxm = sx[m];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + i * 16 + kq * 4;
            if (n >= N) continue;
            const h4 sh = __builtin_bit_cast(h4, *reinterpret_cast<const uint2*>(scale + n));
            store4_i8<EPI>(yv, ldy, m, n, acc[i][j], sxm, sh);
        }
    }
}

// skinny / generic: block = NW waves = NW K slices of 16 weight rows; MT = 16-row activation tiles per block (grid.y walks M)
template <int MT, int EPI, int NW>
__global__ __launch_bounds__(NW * 64) void gemv_i8_kernel(const int8_t* __restrict__ xq, const float* __restrict__ sx,
                                                          const int8_t* __restrict__ w, const uint16_t* __restrict__ scale, int64_t M, int N,
                                                          int K, void* __restrict__ yv, int64_t ldy) {
    __shared__ __attribute__((aligned(16))) int red[NW - 1][MT][64][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int64_t mbase = (int64_t)blockIdx.y * (MT * 16);
    int n = n0 + l15;
    if (n >= N) n = N - 1;
    const int steps = (K + 63) / 64;  // one wave-load = 16 rows x 64 bytes
    const int per = (steps + NW - 1) / NW;
    const int s_begin = wave * per, s_end = (s_begin + per < steps) ? s_begin + per : steps;
    const int8_t* wrow = w + (int64_t)n * K;
    const int8_t* xrow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int64_t m = mbase + mt * 16 + l15;
        if (m >= M) m = M - 1;
        xrow[mt] = xq + m * K;
    }
    i4v acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = i4v{0, 0, 0, 0};
    constexpr int U = 8;  // wave-loads of weights in flight (the block streams its rows once: bytes in flight = bandwidth)
    for (int st0 = s_begin; st0 < s_end; st0 += U) {
        i4v wr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = (st0 + u) * 64 + kq * 16;
            const bool ok = st0 + u < s_end && k < K;  // K % 16 == 0 (launcher)
            wr[u] = ok ? *reinterpret_cast<const i4v*>(wrow + k) : i4v{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = (st0 + u) * 64 + kq * 16;
            const bool ok = st0 + u < s_end && k < K;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const i4v xr = ok ? *reinterpret_cast<const i4v*>(xrow[mt] + k) : i4v{0, 0, 0, 0};
                acc[mt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wr[u], xr, acc[mt], 0, 0, 0);
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) *reinterpret_cast<i4v*>(red[wave - 1][mt][lane]) = acc[mt];
    }
    __syncthreads();
    if (wave == 0) {
        const int nn = n0 + kq * 4;
        if (nn < N) {
            const h4 sh = __builtin_bit_cast(h4, *reinterpret_cast<const uint2*>(scale + nn));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                i4v v = acc[mt];
#pragma unroll
                for (int ww = 0; ww < NW - 1; ++ww) v += *reinterpret_cast<const i4v*>(red[ww][mt][lane]);
                const int64_t m = mbase + mt * 16 + l15;
                if (m >= M) continue;
                store4_i8<EPI>(yv, ldy, m, nn, v, sx[m], sh);
            }
        }
    }
}

hipError_t launch_quant_act(hipStream_t s, const uint16_t* x, int64_t M, int K, int64_t ldx, int8_t* q, int64_t ldq, float* sx) {
    if (M == 0) return hipSuccess;
    if (K % 8 || ldx % 8 || ldq % 8 || ldq < K) return hipErrorInvalidValue;
    hipLaunchKernelGGL(quant_act_kernel, dim3((unsigned)M), dim3(256), 0, s, x, K, ldx, q, ldq, sx);
    return hipGetLastError();
}

hipError_t launch_quant_weight(hipStream_t s, const uint16_t* w, int N, int K, int8_t* q, int64_t ldq, uint16_t* scale) {
    if (N == 0) return hipSuccess;
    if (ldq < K) return hipErrorInvalidValue;
    hipLaunchKernelGGL(quant_weight_kernel, dim3((unsigned)N), dim3(256), 0, s, w, K, q, ldq, scale);
    return hipGetLastError();
}

hipError_t launch_linear_i8(hipStream_t s, const int8_t* xq, const float* sx, const int8_t* w, const uint16_t* scale, int64_t M, int N,
                            int K, void* y, int64_t ldy, bool out_fp32, bool swiglu) {
    if (M == 0) return hipSuccess;
    if (swiglu && out_fp32) return hipErrorInvalidValue;
    if (N % 4 || ldy % 4 || K % 16) return hipErrorInvalidValue;
    const int epi = swiglu ? EPI_SWIGLU : (out_fp32 ? EPI_F32 : EPI_F16);
    static const bool force_generic = tune_set("PPLHIP_GEMM_GENERIC");
    if (M > 32 && K % I_BK == 0 && !force_generic) {
        const int n_tiles = (N + I_BN - 1) / I_BN, m_tiles = (int)((M + I_BM - 1) / I_BM);
        const size_t lds = 4 * (size_t)I_BM * I_BK;
        static bool attr_dev[64] = {false};  // per device (see launch_linear)
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (!attr_dev[dev & 63]) {
            (void)hipFuncSetAttribute((const void*)gemm_i8_kernel<EPI_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute((const void*)gemm_i8_kernel<EPI_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute((const void*)gemm_i8_kernel<EPI_SWIGLU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_dev[dev & 63] = true;
        }
        dim3 grid((unsigned)((n_tiles + 7) / 8 * 8 * m_tiles));
        // 8-wave producer / consumer blocks everywhere (measured at M = 1024 against the 4-wave kernel: wo 32.5 -> 28.6 us and
        // w2 76.0 -> 51.4 us with a 4-stage ring, one block per CU; w13 116.6 -> 106.0 us with two stages, two blocks per CU; wqkv
        // 62.1 vs 63.6 us; M = 2048 layer 502 -> 459 us, M = 8192 equal).  PPLHIP_GEMM_I8_PC = 0 (4-wave kernel) / 2 / 3 / 4 forces a form.
        static const int min_m256 = tune_int("PPLHIP_GEMM_I8_256_MIN_M", 4096);  // measured: M = 4096 layer 888 us (1.87 POP/s) vs 1110, M = 2048 554 vs 459 us
        if (M >= min_m256 && N >= 1024) {
            const int nt2 = (N + 255) / 256, mt2 = (int)((M + 255) / 256);
            static const int st256 = tune_int("PPLHIP_GEMM_I8_256_ST", 4);
            const size_t lds2 = (size_t)(st256 == 3 ? 3 : 4) * 2 * 256 * 64;
            static bool attr2[64] = {false};
            if (!attr2[dev & 63]) {
#define A2(E) do { (void)hipFuncSetAttribute((const void*)gemm_i8_256_kernel<E, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 2 * 256 * 64); \
                   (void)hipFuncSetAttribute((const void*)gemm_i8_256_kernel<E, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * 256 * 64); } while (0)
                A2(EPI_F16); A2(EPI_F32); A2(EPI_SWIGLU);
#undef A2
                attr2[dev & 63] = true;
            }
            static const int forced_gm = tune_int("PPLHIP_GEMM256_GM", 0);
            static const int forced_gn = tune_int("PPLHIP_GEMM256_GN", 0);
            const int nl = (nt2 + 7) / 8;
            int gm = forced_gm > 0 ? forced_gm : 4;
            while (gm > 1 && mt2 % gm) --gm;
            int gn = forced_gn > 0 ? forced_gn : 8;
            if (gn > nl) gn = nl;
            dim3 g2((unsigned)(8 * ((nl + gn - 1) / gn * gn) * mt2));
#define L2(E) do { if (st256 == 3) hipLaunchKernelGGL((gemm_i8_256_kernel<E, 3>), g2, dim3(512), lds2, s, xq, sx, w, scale, M, N, K, y, ldy, nt2, mt2, gn, gm); \
                   else hipLaunchKernelGGL((gemm_i8_256_kernel<E, 4>), g2, dim3(512), lds2, s, xq, sx, w, scale, M, N, K, y, ldy, nt2, mt2, gn, gm); } while (0)
            if (epi == EPI_F32) L2(EPI_F32); else if (epi == EPI_F16) L2(EPI_F16); else L2(EPI_SWIGLU);
#undef L2
            return hipGetLastError();
        }
        {   // 128 x 384 tiles when they fill rounds of 256 one-per-CU blocks (the rule of linear_w8_wide_waves, k_gemm_wide.hip)
            static const int wide = tune_int("PPLHIP_GEMM_I8_WIDE", 1);
            if (wide && M >= 512 && N >= 8192 && linear_w8_wide_waves(M, N) == 12) {
                const int ntw = (N + IW_BN - 1) / IW_BN, mtw = (int)((M + I_BM - 1) / I_BM);
                const size_t ldsw = (size_t)IW_ST * (IW_XB + IW_WB);
                static bool attr_w[64] = {false};
                if (!attr_w[dev & 63]) {
                    (void)hipFuncSetAttribute((const void*)gemm_i8_wide_kernel<EPI_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw);
                    (void)hipFuncSetAttribute((const void*)gemm_i8_wide_kernel<EPI_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw);
                    (void)hipFuncSetAttribute((const void*)gemm_i8_wide_kernel<EPI_SWIGLU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw);
                    attr_w[dev & 63] = true;
                }
                dim3 gw((unsigned)((ntw + 7) / 8 * 8 * mtw));
#define LW(E) hipLaunchKernelGGL((gemm_i8_wide_kernel<E>), gw, dim3((IW_NC + IW_NP) * 64), ldsw, s, xq, sx, w, scale, M, N, K, y, ldy, ntw, mtw)
                if (epi == EPI_F32) LW(EPI_F32); else if (epi == EPI_F16) LW(EPI_F16); else LW(EPI_SWIGLU);
#undef LW
                return hipGetLastError();
            }
        }
        static const int forced_pc = tune_int("PPLHIP_GEMM_I8_PC", -1);
        const int pc = forced_pc >= 0 ? forced_pc : ((int64_t)n_tiles * m_tiles <= 256 ? 4 : 2);
        if (pc == 2 || pc == 3 || pc == 4) {
            const size_t lds_pc = (size_t)pc * 2 * I_BM * I_BK;
            static bool attr_pc[64] = {false};
            if (!attr_pc[dev & 63]) {
#define PCA(E) do { (void)hipFuncSetAttribute((const void*)gemm_i8_pc_kernel<E, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 2 * I_BM * I_BK); \
                    (void)hipFuncSetAttribute((const void*)gemm_i8_pc_kernel<E, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * I_BM * I_BK); \
                    (void)hipFuncSetAttribute((const void*)gemm_i8_pc_kernel<E, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * I_BM * I_BK); } while (0)
                PCA(EPI_F16); PCA(EPI_F32); PCA(EPI_SWIGLU);
#undef PCA
                attr_pc[dev & 63] = true;
            }
#define LPC(E) do { if (pc == 2) hipLaunchKernelGGL((gemm_i8_pc_kernel<E, 2>), grid, dim3(512), lds_pc, s, xq, sx, w, scale, M, N, K, y, ldy, n_tiles, m_tiles); \
                    else if (pc == 3) hipLaunchKernelGGL((gemm_i8_pc_kernel<E, 3>), grid, dim3(512), lds_pc, s, xq, sx, w, scale, M, N, K, y, ldy, n_tiles, m_tiles); \
                    else hipLaunchKernelGGL((gemm_i8_pc_kernel<E, 4>), grid, dim3(512), lds_pc, s, xq, sx, w, scale, M, N, K, y, ldy, n_tiles, m_tiles); } while (0)
            if (epi == EPI_F32) LPC(EPI_F32); else if (epi == EPI_F16) LPC(EPI_F16); else LPC(EPI_SWIGLU);
#undef LPC
            return hipGetLastError();
        }
#define LT(E) hipLaunchKernelGGL((gemm_i8_kernel<E>), grid, dim3(256), lds, s, xq, sx, w, scale, M, N, K, y, ldy, n_tiles, m_tiles)
        if (epi == EPI_F32) LT(EPI_F32); else if (epi == EPI_F16) LT(EPI_F16); else LT(EPI_SWIGLU);
#undef LT
        return hipGetLastError();
    }
    const int nblk = (N + 15) / 16;
    if (M <= 16) {
        dim3 grid((unsigned)nblk, 1);
#define LV(E) do { if (nblk <= 1024) hipLaunchKernelGGL((gemv_i8_kernel<1, E, 8>), grid, dim3(512), 0, s, xq, sx, w, scale, M, N, K, y, ldy); \
                   else hipLaunchKernelGGL((gemv_i8_kernel<1, E, 4>), grid, dim3(256), 0, s, xq, sx, w, scale, M, N, K, y, ldy); } while (0)
        if (epi == EPI_F32) LV(EPI_F32); else if (epi == EPI_F16) LV(EPI_F16); else LV(EPI_SWIGLU);
#undef LV
        return hipGetLastError();
    }
    dim3 grid((unsigned)nblk, (unsigned)((M + 31) / 32));
#define LV(E) hipLaunchKernelGGL((gemv_i8_kernel<2, E, 4>), grid, dim3(256), 0, s, xq, sx, w, scale, M, N, K, y, ldy)
    if (epi == EPI_F32) LV(EPI_F32); else if (epi == EPI_F16) LV(EPI_F16); else LV(EPI_SWIGLU);
#undef LV
    return hipGetLastError();
}

}  // namespace pplhip
