// K3 / K9 / K11 linear layers: y[M,N] = x[M,K] . W[N,K]^T with weight-only quantisation (W8A16 per output
// channel, W4A16 per K-group, or plain fp16 weights).  MFMA-bound at the batch sizes of the headline
// configuration (M = 1024 rows: ~2*M flop per weight byte >> machine balance), so the kernel is an LDS-tiled
// mfma_f32_16x16x32_f16 GEMM with the int8 / int4 weights dequantised to fp16 on their way into LDS.
//
//   block tile 128 (weight rows n) x 128 (activation rows m) x 64 (k); 4 waves as 2 x 2, each 64 x 64 =
//   4 x 4 MFMA tiles; the WEIGHT fragment is the MFMA A operand and the activation fragment the B operand, so
//   the accumulator layout gives each lane 4 consecutive n of one activation row -> 8-byte (fp16) stores and
//   an 8-byte load of the 4 per-channel scales in the epilogue.
//   LDS tiles are [row][64] fp16 with the 16-byte chunk index XOR-swizzled by (row>>1)&7 (128-byte rows: two
//   rows per 256-byte bank window) so ds_read_b128 fragment reads are conflict-free.
//   global -> register -> LDS staging with the next tile's loads in flight during the MFMAs of the current.
//   1-D grid remapped so that the M-tiles of one weight tile run on the same XCD (its L2 then serves the tile
//   to all of them; MI355X_MICROARCH: block b -> XCD b % 8).
// Numerics: fp32 accumulate; W8: y = scale[n] * sum (exact int8 -> fp16); W4: fp16(q * scale) per element
// (one rounding).  Oracle: ref_linear_fwd (oracle/llama_ref.c).
#include <stdlib.h>

#include <algorithm>

#include <type_traits>
#include "k_gemm_dev.h"

namespace pplhip {

template <int WQ, int EPI>
__global__ __launch_bounds__(256) void gemm_kernel(const uint16_t* __restrict__ x, const void* __restrict__ wv,
                                                   const uint16_t* __restrict__ scale, int64_t M, int N, int K, int group,
                                                   void* __restrict__ yv, int64_t ldy, int n_tiles, int m_tiles) {
    __shared__ __attribute__((aligned(16))) uint16_t Ws[G_BN * G_BK];
    __shared__ __attribute__((aligned(16))) uint16_t Xs[G_BM * G_BK];

    // XCD-aware tile mapping: XCD x = id % 8 walks weight tiles n = x, x+8, ... and for each all m tiles
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int nt = xcd + 8 * (slot / m_tiles);
    const int mt = slot % m_tiles;
    if (nt >= n_tiles) return;
    const int n0 = nt * G_BN;
    const int64_t m0 = (int64_t)mt * G_BM;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int wn = wave >> 1, wm = wave & 1;

    f4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    // staging registers
    uint4 xr[4];
    uint4 wr[WQ == 0 ? 4 : (WQ == 8 ? 2 : 1)];
    float wsc = 1.f;  // W4: group scale of this thread's chunk

    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + 256 * i, row = q >> 3, cc = q & 7;
            const int64_t m = m0 + row;
            const int k = k0 + cc * 8;
            xr[i] = (m < M && k < K) ? *reinterpret_cast<const uint4*>(x + m * K + k) : make_uint4(0, 0, 0, 0);
        }
        if constexpr (WQ == 0) {
            const uint16_t* w = reinterpret_cast<const uint16_t*>(wv);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = tid + 256 * i, row = q >> 3, cc = q & 7;
                const int n = n0 + row, k = k0 + cc * 8;
                wr[i] = (n < N && k < K) ? *reinterpret_cast<const uint4*>(w + (int64_t)n * K + k) : make_uint4(0, 0, 0, 0);
            }
        } else if constexpr (WQ == 8) {
            const int8_t* w = reinterpret_cast<const int8_t*>(wv);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = tid + 256 * i, row = q >> 2, c16 = q & 3;
                const int n = n0 + row, k = k0 + c16 * 16;
                wr[i] = (n < N && k < K) ? *reinterpret_cast<const uint4*>(w + (int64_t)n * K + k) : make_uint4(0, 0, 0, 0);
            }
        } else {
            const uint8_t* w = reinterpret_cast<const uint8_t*>(wv);
            const int row = tid >> 1, c32 = tid & 1;
            const int n = n0 + row, k = k0 + c32 * 32;
            const bool ok = n < N && k < K;
            wr[0] = ok ? *reinterpret_cast<const uint4*>(w + ((int64_t)n * K + k) / 2) : make_uint4(0, 0, 0, 0);
            wsc = ok ? h2f(scale[(int64_t)n * (K / group) + k / group]) : 0.f;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + 256 * i, row = q >> 3, cc = q & 7;
            *reinterpret_cast<uint4*>(&Xs[row * G_BK + g_swz(row, cc) * 8]) = xr[i];
        }
        if constexpr (WQ == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = tid + 256 * i, row = q >> 3, cc = q & 7;
                *reinterpret_cast<uint4*>(&Ws[row * G_BK + g_swz(row, cc) * 8]) = wr[i];
            }
        } else if constexpr (WQ == 8) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = tid + 256 * i, row = q >> 2, c16 = q & 3;
                const uint32_t w4[4] = {wr[i].x, wr[i].y, wr[i].z, wr[i].w};
#pragma unroll
                for (int hc = 0; hc < 2; ++hc) {
                    h8 v;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int idx = hc * 8 + e;
                        v[e] = (_Float16)(int)(int8_t)(w4[idx >> 2] >> (8 * (idx & 3)));
                    }
                    *reinterpret_cast<uint4*>(&Ws[row * G_BK + g_swz(row, c16 * 2 + hc) * 8]) = __builtin_bit_cast(uint4, v);
                }
            }
        } else {
            const int row = tid >> 1, c32 = tid & 1;
            const uint32_t w4[4] = {wr[0].x, wr[0].y, wr[0].z, wr[0].w};
#pragma unroll
            for (int hc = 0; hc < 4; ++hc) {
                h8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int idx = hc * 8 + e;  // element 0..31; byte idx>>1, low nibble = even element
                    const uint32_t byte = (w4[idx >> 3] >> (8 * ((idx >> 1) & 3))) & 0xffu;
                    const int nib = (idx & 1) ? (int)(byte >> 4) : (int)(byte & 15u);
                    v[e] = (_Float16)((float)(nib - 8) * wsc);
                }
                *reinterpret_cast<uint4*>(&Ws[row * G_BK + g_swz(row, c32 * 4 + hc) * 8]) = __builtin_bit_cast(uint4, v);
            }
        }
    };

    const int ktiles = (K + G_BK - 1) / G_BK;
    load_tile(0);
    store_tile();
    __syncthreads();
    for (int t = 0; t < ktiles; ++t) {
        if (t + 1 < ktiles) load_tile((t + 1) * G_BK);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            h8 a[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wn * 64 + i * 16 + l15;
                a[i] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(&Ws[row * G_BK + g_swz(row, ks * 4 + kq) * 8]));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wm * 64 + j * 16 + l15;
                bfr[j] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(&Xs[row * G_BK + g_swz(row, ks * 4 + kq) * 8]));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (t + 1 < ktiles) {
            store_tile();
            __syncthreads();
        }
    }

    // epilogue: lane holds, for activation row m = .. + l15, weight rows n = .. + kq*4 + r (r = 0..3)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + wn * 64 + i * 16 + kq * 4;
        if (n >= N) continue;  // N is a multiple of 4 (checked by the launcher)
        float sc[4] = {1.f, 1.f, 1.f, 1.f};
        if constexpr (WQ == 8) {
            const uint2 s2 = *reinterpret_cast<const uint2*>(scale + n);
            const h4 sh = __builtin_bit_cast(h4, s2);
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[r] = (float)sh[r];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t m = m0 + wm * 64 + j * 16 + l15;
            if (m >= M) continue;
            store4<EPI>(yv, ldy, m, n, acc[i][j][0] * sc[0], acc[i][j][1] * sc[1], acc[i][j][2] * sc[2], acc[i][j][3] * sc[3]);
        }
    }
}

// stand-alone launch of the tile body (k_gemm_dev.h)
template <int WQ, int EPI, int G_ST, int WL, int BM = G_BM, int MAXG = W4_MAXG>
__global__ __launch_bounds__(WL == 6 ? 768 : (WL == 5 ? 512 : 256)) void gemm_dma_kernel(const uint16_t* __restrict__ x, const void* __restrict__ wv,
                                                             const uint16_t* __restrict__ scale, int64_t M, int N, int K,
                                                             void* __restrict__ yv, int64_t ldy, int n_tiles, int m_tiles,
                                                             int map_mode, int kt_per_split, float* __restrict__ ws) {
    __shared__ __attribute__((aligned(16))) char smem[gemm_dma_lds_bytes<WQ, G_ST, BM, MAXG>()];
    gemm_dma_body<WQ, EPI, G_ST, WL, BM>(x, wv, scale, M, N, K, yv, ldy, n_tiles, m_tiles, map_mode, kt_per_split, ws, (int)blockIdx.x,
                                     (int)blockIdx.y, (int)gridDim.y, smem);
}

// ---------------------------------------------------------------------------------------------------------------
// W8A16, 4 < M <= 64 (decode steps of a few dozen requests): the half-height tile with a 128-deep K tile (round 4).
// These steps are bound by the weight stream (2 M flop per weight byte), and the 64-deep tile of gemm_dma_body fetches HALF a 128-byte
// line of every weight row per LDS-DMA piece (16 rows x 64 B) -- the access pattern that held the MFMA skinny kernel at 2.75 TB/s
// (k_gemv.hip: whole lines 4.3 TB/s); the 7B layer at M = 64 ran at 2.4 TB/s.  Here a piece is 8 weight rows x 128 B: the weight stage
// is [128 rows][128 B] (two 64-deep sub-tiles side by side, chunk-swizzled like an activation tile), the activation stage two
// [64 rows][64 fp16] sub-tiles; 4 waves of 32 (n) x 64 (m), 32 MFMAs per wave and stage, split-K slabs + the common reduce kernel.
// ---------------------------------------------------------------------------------------------------------------
// Round 4, second step: the activation sub-tile is as tall as the batch needs (BM = 16 / 32 / 64 rows), so that the LDS a block does not
// spend on padding rows holds more stages of the weight stream: what bounds these steps is weight bytes in flight per CU (32 KiB with two
// 64-row blocks of two stages; 64 KiB with three stages of a 16- or 32-row block, two blocks per CU).
constexpr int S_KS = 2;                                             // 64-deep sub-tiles per stage
constexpr int S_WB = G_BN * G_BK * S_KS;                            // weight bytes per stage: 16 KiB
template <int BM> constexpr int s_xb() { return S_KS * BM * G_BK * 2; }  // activation bytes per stage: 4 / 8 / 16 KiB

template <int EPI, int ST, int BM>
__global__ __launch_bounds__(256) void gemm_w8_half128_kernel(const uint16_t* __restrict__ x, const int8_t* __restrict__ w,
                                                              const uint16_t* __restrict__ scale, int64_t M, int N, int K,
                                                              void* __restrict__ yv, int64_t ldy, int n_tiles, int kt_per_split,
                                                              float* __restrict__ ws) {
    constexpr int XB = s_xb<BM>(), NJ = BM / 16;
    constexpr int XP = XB / 1024, WP = S_WB / 1024, PP = (XP + WP) / 4;   // 1-KiB DMA pieces per stage: X, W, per wave (5 / 6 / 8)
    static_assert((XP + WP) % 4 == 0, "every wave issues the same number of pieces");
    extern __shared__ __attribute__((aligned(16))) char smem_h[];  // ST x X, then ST x W
    char* const Xs0 = smem_h;
    char* const Wq0 = smem_h + ST * XB;
    const int nt = blockIdx.x;               // (one activation tile; XCD id % 8 streams the weight tiles n == id (mod 8))
    if (nt >= n_tiles) return;
    const int n0 = nt * G_BN;
    const int split_id = blockIdx.y, n_splits = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int nb = wave * 32;

    // piece P = wave + 4 j of a stage.  P < XP: activation sub-tile u = P / (BM / 8), rows 8 (P % (BM / 8)) .. + 8, chunk permutation of
    // gemm_dma_body (ONE 16-byte read of an int8 row then serves both k-steps of a lane); else weight rows 8 (P - XP) .. + 8, 128 B each
    const char* psrc[PP];
    uint32_t pdst[PP];
    const uint32_t xbase = lds_addr(Xs0), wbase = lds_addr(Wq0);
#pragma unroll
    for (int j = 0; j < PP; ++j) {
        const int P = wave + 4 * j;
        if (P < XP) {
            const int u = P / (BM / 8), q = P % (BM / 8);
            const int row = q * 8 + (lane >> 3), pos = (lane & 7) ^ ((row >> 1) & 7);
            const int c = ((pos & 3) << 1) | (pos >> 2);
            int64_t m = row;
            if (m >= M) m = M - 1;
            psrc[j] = reinterpret_cast<const char*>(x + m * K + u * G_BK + c * 8);
            pdst[j] = __builtin_amdgcn_readfirstlane(xbase + P * 1024);
        } else {
            const int row = (P - XP) * 8 + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
            int n = n0 + row;
            if (n >= N) n = N - 1;
            psrc[j] = reinterpret_cast<const char*>(w) + (int64_t)n * K + c * 16;
            pdst[j] = __builtin_amdgcn_readfirstlane(wbase + (P - XP) * 1024);
        }
    }
    auto issue = [&](int stage, int k0) {
#pragma unroll
        for (int j = 0; j < PP; ++j) {
            const bool isx = wave + 4 * j < XP;   // (wave-uniform)
            glds16(psrc[j] + (isx ? (int64_t)k0 * 2 : (int64_t)k0), pdst[j] + stage * (isx ? XB : S_WB));
        }
    };
    constexpr int D = ST - 1;

    f4 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    const int kt_all = K / (G_BK * S_KS);
    const int kt0 = split_id * kt_per_split;
    const int ktiles = (kt0 + kt_per_split < kt_all) ? kt_per_split : kt_all - kt0;
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < ktiles) issue(d, (kt0 + d) * (G_BK * S_KS));
    int st = 0, stn = D % ST;
    for (int t = 0; t < ktiles; ++t) {
        const int younger = (ktiles - 1 - t) < (D - 1) ? (ktiles - 1 - t) : (D - 1);
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PP) : "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + D < ktiles) issue(stn, (kt0 + t + D) * (G_BK * S_KS));
        const char* xs0 = Xs0 + st * XB;
        const char* wq = Wq0 + st * S_WB;
        st = st == ST - 1 ? 0 : st + 1;
        stn = stn == ST - 1 ? 0 : stn + 1;
#pragma unroll
        for (int u = 0; u < S_KS; ++u) {
            const uint16_t* xs = reinterpret_cast<const uint16_t*>(xs0 + u * (BM * G_BK * 2));
            uint4 wraw[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = nb + i * 16 + l15;
                wraw[i] = *reinterpret_cast<const uint4*>(&wq[row * 128 + g_swz(row, u * 4 + kq) * 16]);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                h8 a[2], bfr[NJ];
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i] = cvt_i8x8_f16(ks == 0 ? make_uint2(wraw[i].x, wraw[i].y) : make_uint2(wraw[i].z, wraw[i].w));
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int row = j * 16 + l15;
                    bfr[j] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(&xs[row * G_BK + g_swz(row, ks * 4 + kq) * 8]));
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    if (n_splits > 1) {  // fp32 partial slab [split][M][N]
        float* slab = ws + (int64_t)split_id * M * N;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int n = n0 + nb + i * 16 + kq * 4;
            if (n >= N) continue;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int64_t m = j * 16 + l15;
                if (m < M) *reinterpret_cast<float4*>(slab + m * N + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int n = n0 + nb + i * 16 + kq * 4;
        if (n >= N) continue;
        const h4 sh = __builtin_bit_cast(h4, *reinterpret_cast<const uint2*>(scale + n));
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int64_t m = j * 16 + l15;
            if (m >= M) continue;
            store4<EPI>(yv, ldy, m, n, acc[i][j][0] * (float)sh[0], acc[i][j][1] * (float)sh[1], acc[i][j][2] * (float)sh[2],
                        acc[i][j][3] * (float)sh[3]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// W8A16, large M (prefill steps, M >= 4096): 256 x 256 x 64 block tile, 8 waves as 8 (n) x 1 (m), wave tile
// 32 (n) x 256 (m) = 2 x 16 MFMA tiles (each weight fragment is converted once and feeds 16 MFMAs; +2.5 % over 4 x 2).  Same DMA / int8-in-LDS / swizzle scheme as the 128 x 128 kernel, but per MFMA it
// reads ~40 % fewer LDS bytes and converts half as many weight fragments, and each activation / weight byte fetched
// from L2 feeds twice the flops.  One block per CU (3-stage ring = 144 KiB LDS), prefetch distance 2.
// ---------------------------------------------------------------------------------------------------------------
constexpr int H_BN = 256, H_BM = 256, H_ST = 3;

template <int EPI>
__global__ __launch_bounds__(512) void gemm_w8_dma256_kernel(const uint16_t* __restrict__ x, const int8_t* __restrict__ w,
                                                             const uint16_t* __restrict__ scale, int64_t M, int N, int K,
                                                             void* __restrict__ yv, int64_t ldy, int n_tiles, int m_tiles, int gn, int gm, int staged) {
    extern __shared__ __attribute__((aligned(16))) char smem256[];  // H_ST x (X 32 KiB + W 16 KiB)
    uint16_t* const Xs0 = reinterpret_cast<uint16_t*>(smem256);
    int8_t* const Wq0 = reinterpret_cast<int8_t*>(smem256 + H_ST * H_BM * G_BK * 2);

    // block -> tile: XCD x = id % 8 owns the weight tiles n == x (mod 8) and walks them in super-tiles of gn (n) x gm (m) tiles, m
    // fastest inside a super-tile, super-tiles m-major: the ~32 blocks an XCD runs at a time then touch gn weight tiles and gm
    // activation tiles per K step instead of 1 and 32 (gn = 1, gm = m_tiles: the old order), i.e. a third of the bytes through its L2
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int per_super = gn * gm, sm_count = m_tiles / gm;
    const int sup = slot / per_super, within = slot % per_super;
    const int nt = xcd + 8 * ((sup / sm_count) * gn + within / gm);
    const int mt = (sup % sm_count) * gm + within % gm;
    if (nt >= n_tiles) return;
    const int n0 = nt * H_BN;
    const int64_t m0 = (int64_t)mt * H_BM;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int wn = wave, wm = 0;  // 8 (n) x 1 (m): wave tile 32 (n) x 256 (m); every weight fragment is converted by one wave only

    const uint16_t* xsrc[4];
    const int8_t* wsrc[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = j * 512 + tid, row = p >> 3, pos = (p & 7) ^ ((row >> 1) & 7);
        const int c = ((pos & 3) << 1) | (pos >> 2);
        int64_t m = m0 + row;
        if (m >= M) m = M - 1;
        xsrc[j] = x + m * K + c * 8;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = j * 512 + tid, row = p >> 2, c = (p & 3) ^ w_swz(row);
        int n = n0 + row;
        if (n >= N) n = N - 1;
        wsrc[j] = w + (int64_t)n * K + c * 16;
    }
    const uint32_t xdst = __builtin_amdgcn_readfirstlane(lds_addr(Xs0) + wave * 1024);
    const uint32_t wdst = __builtin_amdgcn_readfirstlane(lds_addr(Wq0) + wave * 1024);
    auto issue = [&](int stage, int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(xsrc[j] + k0, xdst + stage * (H_BM * G_BK * 2) + j * 8192);
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(wsrc[j] + k0, wdst + stage * (H_BN * G_BK) + j * 8192);
    };

    f4 acc[2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    const int ktiles = K / G_BK;
    issue(0, 0);
    if (ktiles > 1) issue(1, G_BK);
    int st = 0, stn = 2;
    for (int t = 0; t < ktiles; ++t) {
        if (t + 1 < ktiles) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 2 < ktiles) issue(stn, (t + 2) * G_BK);
        const uint16_t* xs = Xs0 + st * (H_BM * G_BK);
        const int8_t* wq = Wq0 + st * (H_BN * G_BK);
        st = st == H_ST - 1 ? 0 : st + 1;
        stn = stn == H_ST - 1 ? 0 : stn + 1;
        uint4 wraw[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wn * 32 + i * 16 + l15;
            wraw[i] = *reinterpret_cast<const uint4*>(&wq[row * G_BK + (kq ^ w_swz(row)) * 16]);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            h8 a[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = cvt_i8x8_f16(ks == 0 ? make_uint2(wraw[i].x, wraw[i].y) : make_uint2(wraw[i].z, wraw[i].w));
#pragma unroll
            for (int jh = 0; jh < 4; ++jh) {
                h8 bfr[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = wm * 128 + (jh * 4 + j) * 16 + l15;
                    bfr[j] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(&xs[row * G_BK + g_swz(row, ks * 4 + kq) * 8]));
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][jh * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], bfr[j], acc[i][jh * 4 + j], 0, 0, 0);
            }
        }
    }

    if (staged && EPI != EPI_F32) {
        // Output through LDS (the ring is idle; round 4): the MFMA layout gives a store instruction 32 bytes of 16 different rows; staged,
        // a wave stores 512 contiguous bytes of two rows (16 bytes per lane).  On multi-round prefill GEMMs the partial-line stores of
        // finishing blocks otherwise sit in the memory pipeline beside the running blocks' tile fetches (measured on the asm-loop
        // kernel: w13 at M = 8192 1593 -> 1382 us, profiles/r04_gemm_asm_experiments.md).  Rows padded by 16 bytes against bank conflicts.
        constexpr int OUTW = EPI == EPI_SWIGLU ? H_BN / 2 : H_BN, ROWB = OUTW * 2 + 16, CPR = OUTW / 8;
        static_assert(H_BM * ROWB <= H_ST * (H_BM * G_BK * 2 + H_BN * G_BK), "the staging image fits the ring");
        char* const stg = smem256;
        __syncthreads();  // every wave is past its last fragment read
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int nl = wn * 32 + i * 16 + kq * 4;
            const int nc = n0 + nl < N ? n0 + nl : 0;
            const h4 sh = __builtin_bit_cast(h4, *reinterpret_cast<const uint2*>(scale + nc));
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float v0 = acc[i][j][0] * (float)sh[0], v1 = acc[i][j][1] * (float)sh[1], v2 = acc[i][j][2] * (float)sh[2],
                            v3 = acc[i][j][3] * (float)sh[3];
                char* dst = stg + (j * 16 + l15) * ROWB;
                if constexpr (EPI == EPI_F16) {
                    const h4 o = {to_h(v0), to_h(v1), to_h(v2), to_h(v3)};
                    *reinterpret_cast<uint2*>(dst + nl * 2) = __builtin_bit_cast(uint2, o);
                } else {
                    const float g0 = round_h(v0), u0 = round_h(v1), g1 = round_h(v2), u1 = round_h(v3);
                    const h2 o = {to_h(g0 / (1.0f + __expf(-g0)) * u0), to_h(g1 / (1.0f + __expf(-g1)) * u1)};
                    *reinterpret_cast<uint32_t*>(dst + nl) = __builtin_bit_cast(uint32_t, o);
                }
            }
        }
        __syncthreads();
        uint16_t* const y = reinterpret_cast<uint16_t*>(yv);
        const int nout = EPI == EPI_SWIGLU ? N / 2 : N, c0 = EPI == EPI_SWIGLU ? n0 / 2 : n0;
#pragma unroll 4
        for (int c = tid; c < H_BM * CPR; c += 512) {
            const int row = c / CPR, ch = c - row * CPR;
            const uint4 v = *reinterpret_cast<const uint4*>(stg + row * ROWB + ch * 16);
            const int64_t m = m0 + row;
            const int col = c0 + ch * 8;
            if (m < M) {
                if (col + 8 <= nout) {
                    *reinterpret_cast<uint4*>(y + m * ldy + col) = v;
                } else {
                    const uint16_t* e = reinterpret_cast<const uint16_t*>(&v);
                    for (int k = 0; k < 8; ++k)
                        if (col + k < nout) y[m * ldy + col + k] = e[k];
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int n = n0 + wn * 32 + i * 16 + kq * 4;
        if (n >= N) continue;
        const h4 sh = __builtin_bit_cast(h4, *reinterpret_cast<const uint2*>(scale + n));
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int64_t m = m0 + wm * 128 + j * 16 + l15;
            if (m >= M) continue;
            store4<EPI>(yv, ldy, m, n, acc[i][j][0] * (float)sh[0], acc[i][j][1] * (float)sh[1], acc[i][j][2] * (float)sh[2],
                        acc[i][j][3] * (float)sh[3]);
        }
    }
}

template <int WQ, int MT, int EPI, int NW>
__global__ __launch_bounds__(NW * 64) void gemv_kernel(const uint16_t* __restrict__ x, const void* __restrict__ wv,
                                                   const uint16_t* __restrict__ scale, int64_t M, int N, int K, int group,
                                                   void* __restrict__ yv, int64_t ldy) {
    constexpr int KL = WQ == 8 ? 16 : (WQ == 4 ? 32 : 8);  // k elements in one lane's 16 bytes
    constexpr int KSTEP = KL * 4;                          // k elements per wave-load
    constexpr int NKS = KL / 8;                            // MFMA k-steps per load
    __shared__ __attribute__((aligned(16))) float red[NW - 1][MT][64][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int n0 = blockIdx.x * 16;
    int n = n0 + l15;
    if (n >= N) n = N - 1;
    const int steps = (K + KSTEP - 1) / KSTEP;
    const int per = (steps + NW - 1) / NW;
    const int s_begin = wave * per, s_end = (s_begin + per < steps) ? s_begin + per : steps;
    const char* wrow = reinterpret_cast<const char*>(wv) + ((int64_t)n * K * (WQ == 0 ? 16 : WQ)) / 8;
    const uint16_t* xrow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int64_t m = mt * 16 + l15;
        if (m >= M) m = M - 1;
        xrow[mt] = x + m * K;
    }
    f4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f4{0.f, 0.f, 0.f, 0.f};

    // U wave-loads of weights are issued before the first is consumed: a block streams 16 rows once, so the bytes it keeps
    // in flight (U KiB per wave) set its share of the HBM bandwidth (measured: 2.6 TB/s with one load in flight)
    constexpr int U = 8;
    for (int st0 = s_begin; st0 < s_end; st0 += U) {
        uint4 wr[U];
        float wsc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = (st0 + u) * KSTEP + kq * KL;
            const bool ok = st0 + u < s_end && k < K;  // K is a multiple of KL (checked by the launcher)
            wr[u] = ok ? *reinterpret_cast<const uint4*>(wrow + ((int64_t)k * (WQ == 0 ? 16 : WQ)) / 8) : make_uint4(0, 0, 0, 0);
            wsc[u] = 1.f;
            if constexpr (WQ == 4) wsc[u] = ok ? h2f(scale[(int64_t)n * (K / group) + k / group]) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = (st0 + u) * KSTEP + kq * KL;
            const bool ok = st0 + u < s_end && k < K;
            const uint32_t w4[4] = {wr[u].x, wr[u].y, wr[u].z, wr[u].w};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                h8 a;
                if constexpr (WQ == 0) {
                    a = __builtin_bit_cast(h8, wr[u]);
                } else if constexpr (WQ == 8) {
                    a = cvt_i8x8_f16(make_uint2(w4[ks * 2], w4[ks * 2 + 1]));
                } else {
                    const _Float16 sh = (_Float16)wsc[u];  // exact: wsc came from an fp16
                    a = cvt_i4x8_f16(w4[ks], h2{sh, sh});  // 8 nibbles, low nibble = even k
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const uint4 xr = ok ? *reinterpret_cast<const uint4*>(xrow[mt] + k + ks * 8) : make_uint4(0, 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(h8, xr), acc[mt], 0, 0, 0);
                }
            }
        }
    }
    // sum the NW K slices
    if (wave > 0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) *reinterpret_cast<f4*>(red[wave - 1][mt][lane]) = acc[mt];
    }
    __syncthreads();
    if (wave == 0) {
        const int nn = n0 + kq * 4;
        if (nn < N) {
            float sc[4] = {1.f, 1.f, 1.f, 1.f};
            if constexpr (WQ == 8) {
                const h4 sh = __builtin_bit_cast(h4, *reinterpret_cast<const uint2*>(scale + nn));
#pragma unroll
                for (int r = 0; r < 4; ++r) sc[r] = (float)sh[r];
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f4 v = acc[mt];
#pragma unroll
                for (int w = 0; w < NW - 1; ++w) {
                    const f4 o = *reinterpret_cast<const f4*>(red[w][mt][lane]);
                    v += o;
                }
                const int64_t m = mt * 16 + l15;
                if (m >= M) continue;
                store4<EPI>(yv, ldy, m, nn, v[0] * sc[0], v[1] * sc[1], v[2] * sc[2], v[3] * sc[3]);
            }
        }
    }
}

template <int WQ, int EPI>
static hipError_t launch_gemv(hipStream_t s, const uint16_t* x, const void* w, const uint16_t* scale, int group, int64_t M,
                              int N, int K, void* y, int64_t ldy) {
    dim3 grid((unsigned)((N + 15) / 16));
    const int mt = (int)((M + 15) / 16);
    static const int forced_nw = tune_int("PPLHIP_GEMV_WAVES", 0);
    // up to 1024 row tiles -> 8 K slices per tile so that every CU has enough waves streaming (profiles/gemm_microbench.py)
    const int nw = forced_nw ? forced_nw : ((N + 15) / 16 <= 1024 ? 8 : 4);
#define GEMV_CASE(MT)                                                                                                    \
    if (mt == MT) {                                                                                                      \
        if (nw == 8) hipLaunchKernelGGL((gemv_kernel<WQ, MT, EPI, 8>), grid, dim3(512), 0, s, x, w, scale, M, N, K, group, y, ldy); \
        else hipLaunchKernelGGL((gemv_kernel<WQ, MT, EPI, 4>), grid, dim3(256), 0, s, x, w, scale, M, N, K, group, y, ldy);       \
        return hipGetLastError();                                                                                        \
    }
    GEMV_CASE(1) GEMV_CASE(2)
#undef GEMV_CASE
    return hipErrorInvalidValue;
}

// split-K reduce: y[m][n] = (sum_z slab[z][m][n]) * scale[n]  (scale == NULL: 1)
template <int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int splits, int64_t M, int N,
                                                            const uint16_t* __restrict__ scale, void* __restrict__ yv, int64_t ldy) {
    const int64_t total = M * (N / 4);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / (N / 4);
        const int n = (int)(i - m * (N / 4)) * 4;
        float4 v = *reinterpret_cast<const float4*>(ws + m * N + n);
        for (int z = 1; z < splits; ++z) {
            const float4 o = *reinterpret_cast<const float4*>(ws + ((int64_t)z * M + m) * N + n);
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        if (scale) {
            const h4 sh = __builtin_bit_cast(h4, *reinterpret_cast<const uint2*>(scale + n));
            v.x *= (float)sh[0]; v.y *= (float)sh[1]; v.z *= (float)sh[2]; v.w *= (float)sh[3];
        }
        store4<EPI>(yv, ldy, m, n, v.x, v.y, v.z, v.w);
    }
}

hipError_t launch_splitk_reduce(hipStream_t s, const float* ws, int splits, int64_t M, int N, const uint16_t* scale, void* y, int64_t ldy, int epi) {
    const int64_t total = M * (N / 4);
    if (total == 0) return hipSuccess;
    const unsigned rb = (unsigned)std::min<int64_t>((total + 255) / 256, 2048);
#define RED(E) hipLaunchKernelGGL(splitk_reduce_kernel<E>, dim3(rb), dim3(256), 0, s, ws, splits, M, N, scale, y, ldy)
    if (epi == EPI_F32) RED(EPI_F32); else if (epi == EPI_F16) RED(EPI_F16); else RED(EPI_SWIGLU);
#undef RED
    return hipGetLastError();
}

hipError_t launch_linear(hipStream_t s, const uint16_t* x, const void* w, const uint16_t* scale, int wq_bit, int group,
                         int64_t M, int N, int K, void* y, int64_t ldy, bool out_fp32, float* ws, size_t ws_bytes, bool swiglu, SplitSlabs* defer) {
    if (defer) *defer = SplitSlabs{};
    if (M == 0) return hipSuccess;
    if (swiglu && out_fp32) return hipErrorInvalidValue;
    const int epi = swiglu ? EPI_SWIGLU : (out_fp32 ? EPI_F32 : EPI_F16);
    if (N % 4 || ldy % 4) return hipErrorInvalidValue;
    if (wq_bit == 0 && K % 8) return hipErrorInvalidValue;
    if (wq_bit == 8 && K % 16) return hipErrorInvalidValue;
    if (wq_bit == 4 && (K % 32 || group % 32 || K % group)) return hipErrorInvalidValue;
    // A few rows above a multiple of 1024 (a saturated continuous-batching step: 1024 decode rows + the prompt chunk of a newly admitted
    // request) would add a ninth / seventeenth 128-row tile to every column block -- a whole extra round of tiles for a few rows (7B layer
    // GEMMs at M = 1040: 555 us against 414 us at M = 1024).  The rows up to the multiple go through the shapes that fit, the rest as a
    // second, small launch.
    static const int msplit = tune_int("PPLHIP_GEMM_MSPLIT", 256);  // largest rest that is split off (0: never); measured: rest 16 / 76 / 128 / 256 -13 / -10 / -11 / -5 %, 384 equal
    if (msplit && M > 1024 && M < 3584 && (M & 1023) != 0 && (M & 1023) <= msplit) {
        const int64_t m_main = M & ~(int64_t)1023, m_rest = M - m_main;
        const size_t yelt = out_fp32 ? 4 : 2;
        hipError_t e = launch_linear(s, x, w, scale, wq_bit, group, m_main, N, K, y, ldy, out_fp32, ws, ws_bytes, swiglu);
        if (e != hipSuccess) return e;
        return launch_linear(s, x + m_main * K, w, scale, wq_bit, group, m_rest, N, K, (char*)y + (size_t)m_main * ldy * yelt, ldy, out_fp32, ws,
                             ws_bytes, swiglu);
    }
    static const bool no_skinny = tune_set("PPLHIP_GEMM_NOSKINNY"), force_generic = tune_set("PPLHIP_GEMM_GENERIC");
    // the skinny kernel up to 3 rows; from 4 rows the half-height tile kernel (64 activation rows, split-K) is faster: 7B decode step at batch
    // 4 / 8 / 12 / 16 3.64 / 3.90 / 4.40 / 4.74 -> 3.56 / 3.73 / 3.81 / 4.18 ms (profiles/small_batch_latency.py); PPLHIP_GEMV_MAX_M overrides
    static const int gemv_max_m = tune_int("PPLHIP_GEMV_MAX_M", 3);
    // up to 4 rows (PPLHIP_GEMV_STREAM_MAX_M): the streaming GEMV of k_gemv.hip -- whole 1-KiB row pieces per wave-load, no matrix unit
    if (M <= gemv_stream_max_m(wq_bit, group, N, K) && !no_skinny && !force_generic)
        return launch_gemv_stream(s, x, w, scale, wq_bit, group, M, N, K, y, ldy, epi);
    // (without a split-K workspace the half-height tiles would run as N / 128 unsplit blocks: the skinny kernel keeps its 16 rows there)
    // (int8 weights with K % 128 == 0: from 3 rows the half-height tiles with 16-row activation sub-tiles are faster than either GEMV)
    const int skinny_max = wq_bit == 8 && K % (G_BK * 2) == 0 && gemv_max_m > 2 && !tune_set("PPLHIP_GEMV_MAX_M") ? 2 : gemv_max_m;
    if (M <= (ws && ws_bytes ? skinny_max : 16) && M <= 16 && !no_skinny) {
#define GEMV_DISPATCH(WQ)                                                                                      \
    if (wq_bit == WQ)                                                                                          \
        return epi == EPI_F32 ? launch_gemv<WQ, EPI_F32>(s, x, w, scale, group, M, N, K, y, ldy)               \
             : epi == EPI_F16 ? launch_gemv<WQ, EPI_F16>(s, x, w, scale, group, M, N, K, y, ldy)               \
                              : launch_gemv<WQ, EPI_SWIGLU>(s, x, w, scale, group, M, N, K, y, ldy);
        GEMV_DISPATCH(0) GEMV_DISPATCH(8) GEMV_DISPATCH(4)
#undef GEMV_DISPATCH
    }
    // W4A16 at a few hundred rows (config 4: the 70B / TP8 slice at batch 256): 128 x 64 tiles, DMA waves beside MFMA waves whose fragment
    // registers roll (k_gemm_pc.hip, round 5) -- for the shapes whose 128 x 64 tiles fill most CUs WITHOUT K slabs (wo / w13 / w2 of that
    // slice: 256 / 224 / 256 blocks).  A shape that needs slabs anyway (wqkv: 40 tiles) stays on the ring kernel below: measured in the
    // step, 12.70 -> 12.49 ms (profiles/r05_w4_pc_experiments.md).  PPLHIP_GEMM_PC=0: the ring kernel for everything (A/B runs)
    static const int pc_mode = getenv("PPLHIP_GEMM_PC") ? atoi(getenv("PPLHIP_GEMM_PC")) : 1;
    if (pc_mode && wq_bit == 4 && M > 128 && M <= 512 && !force_generic && (int64_t)((N + 63) / 64) * ((M + 127) / 128) >= 160 &&
        linear_w4_pc_supported(group, M, N, K, x, w, scale, y, ldy, epi))
        return launch_linear_w4_pc(s, x, w, scale, M, N, K, y, ldy, epi);
    // W8A16 at 512 <= M < 4096 with N >= 8192 (wqkv / w13 of a decode step at batch ~1024): one 128 x 384 block per CU whose twelve
    // consumer waves share the activation tile (k_gemm_wide.hip) when its tiles fill the chip's rounds; PPLHIP_GEMM_WIDE=0: never
    static const int wide = tune_int("PPLHIP_GEMM_WIDE", 1);
    if (wide && wq_bit == 8 && K % G_BK == 0 && M >= 512 && (M < 4096 || wide == 2) && N >= 8192) {
        const int nc = wide == 2 ? 12 : linear_w8_wide_waves(M, N);  // (2: experiments -- every eligible shape, any M)
        // (round 4's hand-scheduled K loop for this block tile -- equal speed, both on the chip's power limit -- lives under profiles/probes/gemm_asm/ since round 6)
        if (nc) return launch_linear_w8_wide(s, x, (const int8_t*)w, scale, M, N, K, y, ldy, epi, nc);
    }
    // (Round 6, measured and not adopted: 128 x 96 / 64 x 96 tiles whose four multiplying waves split each K tile by k-step on 32 x 32 x 16
    // MFMAs -- 232 / 256 blocks instead of 176 / 96 for the 7B / TP8 slice's w13 / wqkv, 22 KiB of fragment reads per K tile instead of 72.
    // Parity-green, and no faster: w13 35.5 against 34.9 us, the emulated TP-8 step +0.07 ms.  Its ablation builds say why -- with the MFMAs
    // compiled out the K loop still takes 0.29 us per tile: the 22-24 KiB of LDS-DMA per K tile and CU move at ~40 B / clk whatever the tile
    // computes, and ~9 us of every launch are dispatch, first-byte latency, the final exchange and the stores.
    // profiles/probes/gemm_ks/, profiles/r06_tp_slice_gemm_experiments.md)
    const int n_tiles = (N + G_BN - 1) / G_BN;
    const int m_tiles = (int)((M + G_BM - 1) / G_BM);
    const int n_tiles_pad = (n_tiles + 7) / 8 * 8;
    dim3 grid((unsigned)(n_tiles_pad * m_tiles)), block(256);
    static const int min_m256 = tune_int("PPLHIP_GEMM_256_MIN_M", 3584);  // measured (7B layer): M = 3584 1435 vs 1501 us, 3072 1277 vs 1264, 2560 equal, 2048 941 vs 851
    if (wq_bit == 8 && K % G_BK == 0 && M >= min_m256 && N >= 1024 && !force_generic) {
        // (Round 6, measured and not adopted: the same block tile on v_mfma_f32_32x32x16_f16 -- four waves of 64 (n) x 256 (m) with 256 resident
        // accumulators, or this kernel's eight waves of 32 x 256 -- with two rolling fragment sets and LDS-DMA pieces on a scalar base: a third
        // fewer instructions per flop, parity-green, and 13 % SLOWER (1.05-1.06 against 1.21 PFLOP/s at M = 8192, profiles/r06_gemm_big_ab.log).
        // The counters that motivated it -- 2.25 other instructions per MFMA here against 0.72 in the vendor library's kernel, same L2 hit rates,
        // same bytes -- are in profiles/r06_gemm_bigm_counters.md; the kernel is kept under profiles/probes/gemm_big/.)
        const int nt2 = (N + H_BN - 1) / H_BN, mt2 = (int)((M + H_BM - 1) / H_BM);
        const size_t lds = (size_t)H_ST * (H_BM * G_BK * 2 + H_BN * G_BK);
        // super-tile shape: gm = the largest divisor of mt2 <= 4 (X tiles carry twice the bytes of W tiles), gn = up to 8 weight tiles
        // of the XCD's share; the share is padded to whole super-tiles (the extra blocks return at once)
        static const int forced_gm = tune_int("PPLHIP_GEMM256_GM", 0);
        static const int forced_gn = tune_int("PPLHIP_GEMM256_GN", 0);
        const int nl = (nt2 + 7) / 8;  // weight tiles per XCD
        int gm = 4;
        if (forced_gm > 0) gm = forced_gm;
        while (gm > 1 && mt2 % gm) --gm;
        if (mt2 % gm) gm = 1;
        int gn = forced_gn > 0 ? forced_gn : 8;
        if (gn > nl) gn = nl;
        const int nl_pad = (nl + gn - 1) / gn * gn;
        dim3 g256((unsigned)(8 * nl_pad * mt2));
        // (the attribute belongs to the function ON THE CURRENT DEVICE: one flag per device, or the other ranks of a single-process
        // tensor-parallel run would launch without it)
        static bool attr_dev[64] = {false};
        int dev = 0;
        (void)hipGetDevice(&dev);
        bool& attr_set = attr_dev[dev & 63];
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)gemm_w8_dma256_kernel<EPI_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute((const void*)gemm_w8_dma256_kernel<EPI_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute((const void*)gemm_w8_dma256_kernel<EPI_SWIGLU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_set = true;
        }
        static const int no_stage = tune_set("PPLHIP_GEMM_DIRECT_EPILOGUE") ? 1 : 0;  // A/B runs
        const int staged = !no_stage && ldy % 8 == 0 && ((uintptr_t)y & 15) == 0 && (epi != EPI_SWIGLU || N % 16 == 0) ? 1 : 0;
#define L256(E) hipLaunchKernelGGL((gemm_w8_dma256_kernel<E>), g256, dim3(512), lds, s, x, (const int8_t*)w, scale, M, N, K, y, ldy, nt2, mt2, gn, gm, staged)
        if (epi == EPI_F32) L256(EPI_F32); else if (epi == EPI_F16) L256(EPI_F16); else L256(EPI_SWIGLU);
#undef L256
        return hipGetLastError();
    }
    // W4: group 128 = two K tiles; one block keeps at most W4_MAXG groups of scales in LDS
    const bool w4_fast = wq_bit == 4 && group == 128 && K % 128 == 0 &&
                         (K / 128 <= W4_MAXG || (ws && (size_t)((K / 128 + W4_MAXG - 1) / W4_MAXG) * M * N * sizeof(float) <= ws_bytes));
    // (Round 4, measured and removed: W4 at 128 < M <= 256 as ONE 256-row tile per weight tile -- 4 waves of 32 (n) x 256 (m), every int4
    // fragment converted once for 16 MFMAs instead of 8 -- is SLOWER on the 70B / TP8 shapes: w13 49.1 -> 53.4 us, w2 31.5 -> 38.9 us at
    // M = 256; 236 VGPRs and 88 KiB of LDS leave one wave per SIMD.  profiles/r04_w4_m256_sweep.log)
    if ((wq_bit == 8 || wq_bit == 0 || w4_fast) && K % G_BK == 0 && !force_generic) {
        static const int forced = tune_int("PPLHIP_GEMM_MAP", -1);
        // measured (profiles/gemm_microbench.py, M = 1024): weight tiles per XCD (mode 0) beats activation slices per XCD
        // (mode 1) by 2-4 % on every layer shape
        int map_mode = (forced == 1 && m_tiles % 8 == 0) ? 1 : 0;
        dim3 g2 = map_mode == 1 ? dim3((unsigned)(n_tiles * m_tiles)) : grid;
        static const bool force_super = tune_set("PPLHIP_GEMM128_GM");  // experiments at M <= 1024
        if (map_mode == 0 && (m_tiles > 8 || force_super)) {
            // more than 8 activation tiles (M > 1024): walk the XCD's weight tiles in super-tiles of 12 (n) x 8 (m) -- at M = 1024 the
            // plain order already is that shape; at M = 8192 it degenerates to 1.5 weight tiles x 64 activation tiles in flight per XCD
            static const int forced_gm = tune_int("PPLHIP_GEMM128_GM", 0);
            static const int forced_gn = tune_int("PPLHIP_GEMM128_GN", 0);
            int gm = forced_gm > 0 ? forced_gm : 8;
            while (gm > 1 && m_tiles % gm) --gm;
            const int nl = n_tiles_pad / 8;
            int gn = forced_gn > 0 ? forced_gn : 12;
            if (gn > nl) gn = nl;
            if (gm > 1 && gn <= 255 && gm <= 255) {
                map_mode |= (gm << 16) | (gn << 24);
                g2 = dim3((unsigned)(8 * ((nl + gn - 1) / gn * gn) * m_tiles));
            }
        }
        static const int ablate = tune_int("PPLHIP_GEMM_ABLATE", 0);  // diagnosis only: wrong results
        map_mode |= ablate << 8;
        static const int forced_st = tune_int("PPLHIP_GEMM_STAGES", 0);
        // few blocks per CU -> deeper ring (latency is hidden inside the block); many -> more blocks per CU
        int stages = (int64_t)n_tiles * m_tiles <= 256 ? 4 : 2;  // measured: profiles/gemm_microbench.py
        if (forced_st >= 2 && forced_st <= 4) stages = forced_st;
        if (wq_bit == 0 && stages == 4) stages = 3;  // fp16 weights: 4 x 32 KiB would leave one block per CU anyway
        // split-K for small M: too few output tiles to pull HBM bandwidth (weights must stream at full rate)
        const int kt_all = K / G_BK;
        int splits = 1;
        static const int forced_split = tune_int("PPLHIP_GEMM_SPLITK", 0);
        const int64_t tiles = (int64_t)n_tiles * m_tiles;
        // (also at larger M when a tensor-parallel slice leaves fewer output tiles than CUs)
        if (ws && ((M <= 256 && tiles < 512) || tiles < 200 || forced_split > 0)) {
            // enough blocks to fill the chip, but every split writes an fp32 slab of the whole output: keep >= min_kt K
            // tiles per split (measured at M = 64 / 256 on the 70B/TP8 shapes and M = 1024 on 7B/TP8 slices)
            static const int env_minkt = tune_int("PPLHIP_GEMM_MINKT", 0);
            const int target = M <= 64 ? 768 : 512;
            splits = (int)((target + tiles - 1) / tiles);
            const int cap = (M <= 64 || tiles <= 32) ? 8 : 4;  // (M = 128..256 sweeps: more than 4 slabs never paid)
            if (splits > cap) splits = cap;
            if (M > 64 && tiles >= 160) splits = 1;             // most CUs busy already: a slab costs more (176 tiles, 7B/TP8 w13: 36 vs 42 us)
            const int min_kt = env_minkt ? env_minkt : ((M <= 64 || tiles <= 64) ? 16 : 28);
            if (splits > kt_all / min_kt) splits = kt_all / min_kt > 0 ? kt_all / min_kt : 1;
            if (forced_split > 0) splits = forced_split;
            while (splits > 1 && (size_t)splits * M * N * sizeof(float) > ws_bytes) --splits;
        }
        if (wq_bit == 4 && splits < (kt_all + 2 * W4_MAXG - 1) / (2 * W4_MAXG)) splits = (kt_all + 2 * W4_MAXG - 1) / (2 * W4_MAXG);
        int kt_per = (kt_all + splits - 1) / splits;
        if (wq_bit == 4) kt_per = (kt_per + 1) & ~1;  // whole quantisation groups per split
        splits = (kt_all + kt_per - 1) / kt_per;  // no empty split
        // K slabs: two stages when several blocks share a CU; a launch of at most one block per CU keeps the deep ring (the LDS is free anyway):
        // W8 7B at M = 256 wo 21.3 -> 19.8 us, w2 38.8 -> 33.4; 13B / TP2 at M = 512 wo 28.0 -> 24.3, w2 56.8 -> 52.1; fp16 70B / TP8 w2 30.0 ->
        // 26.0; W4 70B / TP8 w2 31.1 -> 30.1 (profiles/r04_splitk_stages_sweep.log).  Until the last session of round 4 this line forced two
        // stages on EVERY split launch and overrode PPLHIP_GEMM_STAGES, so the earlier stage sweeps never ran 3 / 4 stages on split shapes.
        static const int split_deep = tune_int("PPLHIP_GEMM_SPLIT_DEEP", 1);   // 0: always two stages
        // (W4 half-height tiles at M <= 64 are the exception: w2 of the 70B / TP8 slice 19.4 -> 21.0 us with the deep ring.)  In the steps:
        // 7B / TP8 slice at 1024 rows 7.97 -> 7.69 ms, config 4 per rank 13.68 -> 13.54, 7B TP 1 at batch 160-512 -0.3..-1.9 % (r04_split_deep_ab.log)
        if (splits > 1 && !(forced_st >= 2 && forced_st <= 4))
            stages = (split_deep && tiles * splits <= 256 && !(wq_bit == 4 && M <= 64)) ? (wq_bit == 0 ? 3 : 4) : 2;
        g2.y = splits;
        // at most one block per CU (<= 256 blocks): 8 waves, 4 of them producers that only issue the ring's LDS-DMA (two waves
        // per SIMD from the one block, the DMA issue runs beside the MFMA stream: wo 59 -> 46 us, w2 120 -> 95 us at
        // M = 1024); more blocks: 4 waves of 32(n) x 128(m), two blocks per CU (the specialised form is not faster there)
        static const int forced_wl = tune_int("PPLHIP_GEMM_WL", 0);
        int wl = (forced_wl == 1 || forced_wl == 5) ? forced_wl : (tiles * splits <= 256 ? 5 : 1);
        // 16 < M <= 64: half-height tiles (64 activation rows), 4-wave blocks
        static const int forced_half = tune_int("PPLHIP_GEMM_HALF", -1);
        const bool half = forced_half >= 0 ? (forced_half == 1 && M <= 64) : M <= 64;  // 7B layer GEMMs at M = 64: 102 -> 84 us, M = 32: 90 -> 74 us
        if (half) wl = 1;
        // W8, one block per CU: eight consumer waves, each group multiplying one of the two k-steps of a tile (w2 at M = 1024: 100 -> 96 us)
        if ((forced_wl == 6 || (forced_wl == 0 && wl == 5)) && wq_bit == 8 && stages >= 3 && splits == 1) wl = 6;
        // W8, half-height tiles: the 128-deep K tile (whole 128-byte lines of every weight row per LDS-DMA piece; gemm_w8_half128_kernel),
        // activation sub-tile 16 / 32 / 64 rows; three stages when two blocks of them fit a CU (BM <= 32) or the grid is one block per CU
        static const int half128 = tune_int("PPLHIP_GEMM_HALF128", 1);  // 0: off; 2: always 64-row sub-tiles
        // ... and 64 < M <= 128 with 80- .. 128-row sub-tiles (steps of 16 rows; three stages, one block per CU) for the shapes that need split-K
        // anyway (7B at M = 128, HBM-cold: wqkv 33.1 -> 29.3 us, wo 21.2 -> 18.9, w2 27.9 -> 26.3; w13 -- 172 tiles, no split -- stays on the
        // 128 x 128 ring kernel with its eight consumer waves above 80 rows: 34.7 against 42.0 us at 128).  Decode step at batch 72 / 96 / 128
        // 6.20 / 6.85 / 7.92 -> 5.70 / 6.53 / 7.76 ms.  PPLHIP_GEMM_HALF128_MAX_M=64: off
        static const int half128_max_m = tune_int("PPLHIP_GEMM_HALF128_MAX_M", 128);
        // split-K slabs: ONE block per CU, not three.  A slab costs 8 M N bytes (written here, read by the reduce or the consuming
        // kernel) against N K / splits weight bytes -- at M = 64 and K / splits = 683 that is 0.75 extra bytes per weight byte, and it
        // is written when all blocks finish together.  Measured on the 7B shapes, HBM-cold (profiles/r04_splitk_sweep.log): w13 at
        // M = 64 with 5 / 1 slabs 37.6 / 29.9 us (and no reduce kernel), wqkv 6 / 2 slabs 25.1 / 20.7 us; wo / w2 (32 tiles) keep 8
        static const int h_blocks = tune_int("PPLHIP_GEMM_HALF128_BLOCKS", 256);
        int sp = splits;
        const int kt128 = K / (G_BK * S_KS);
        if (ws && forced_split <= 0) {
            sp = h_blocks / n_tiles;
            if (sp > 8) sp = 8;
            if (sp < 1) sp = 1;
            while (sp > 1 && (size_t)sp * M * N * sizeof(float) > ws_bytes) --sp;
        }
        if (sp > kt128) sp = kt128 > 0 ? kt128 : 1;
        static const int half128_unsplit_max_m = tune_int("PPLHIP_GEMM_HALF128_UNSPLIT_MAX_M", 80);  // (w13: 80-row sub-tile 31.0-31.6 vs 32.4-32.6 us at M = 72-80, 96 rows 34.4 vs 33.1)
        if (half128 && (half || (M <= 128 && M <= half128_max_m && (sp > 1 || M <= half128_unsplit_max_m) && m_tiles == 1)) && wq_bit == 8 && K % (G_BK * S_KS) == 0) {
            const int kt_per128 = (kt128 + sp - 1) / sp;
            sp = (kt128 + kt_per128 - 1) / kt_per128;
            // (above 64 rows in steps of 16: a 72-row step on a 128-row sub-tile would pay 128 rows of LDS traffic and MFMAs)
            const int bm = M > 64 ? (int)((M + 15) / 16 * 16) : (half128 == 2 ? 64 : (M <= 16 ? 16 : (M <= 32 ? 32 : 64)));
            const int st = (bm <= 32 || bm > 64 || (int64_t)n_tiles * sp <= 256) ? 3 : 2;
            const size_t lds = (size_t)st * ((size_t)S_KS * bm * G_BK * 2 + S_WB);
            static bool attr_dev[64] = {false};
            int dev = 0;
            (void)hipGetDevice(&dev);
            if (!attr_dev[dev & 63]) {
#define H128_A(E, S, B) (void)hipFuncSetAttribute((const void*)gemm_w8_half128_kernel<E, S, B>, hipFuncAttributeMaxDynamicSharedMemorySize, S * (s_xb<B>() + S_WB))
#define H128_AE(E) H128_A(E, 3, 16); H128_A(E, 3, 32); H128_A(E, 2, 64); H128_A(E, 3, 64); H128_A(E, 3, 80); H128_A(E, 3, 96); H128_A(E, 3, 112); H128_A(E, 3, 128)
                H128_AE(EPI_F16); H128_AE(EPI_F32); H128_AE(EPI_SWIGLU);
#undef H128_AE
#undef H128_A
                attr_dev[dev & 63] = true;
            }
            dim3 gh((unsigned)n_tiles, (unsigned)sp);
#define H128_L(E, S, B) hipLaunchKernelGGL((gemm_w8_half128_kernel<E, S, B>), gh, dim3(256), lds, s, x, (const int8_t*)w, scale, M, N, K, y, ldy, n_tiles, kt_per128, ws)
#define H128_E(E) do { if (bm == 16) H128_L(E, 3, 16); else if (bm == 32) H128_L(E, 3, 32); else if (bm == 80) H128_L(E, 3, 80); else if (bm == 96) H128_L(E, 3, 96); \
                         else if (bm == 112) H128_L(E, 3, 112); else if (bm == 128) H128_L(E, 3, 128); else if (st == 3) H128_L(E, 3, 64); else H128_L(E, 2, 64); } while (0)
            if (epi == EPI_F32) H128_E(EPI_F32); else if (epi == EPI_F16) H128_E(EPI_F16); else H128_E(EPI_SWIGLU);
#undef H128_E
#undef H128_L
            hipError_t e = hipGetLastError();
            if (e != hipSuccess || sp == 1) return e;
            if (defer && epi == EPI_F16 && N % 8 == 0 && sp <= 8) { *defer = SplitSlabs{ws, sp, scale, N, M}; return e; }  // the consumer reduces
            return launch_splitk_reduce(s, ws, sp, M, N, scale, y, ldy, epi);
        }
        // W4 with K slabs of at most 32 tiles (16 quantisation groups): the scale area shrinks from 16 to 4 KiB and three blocks fit a CU
        static const int w4_sc16 = tune_int("PPLHIP_GEMM_W4_SC16", 1);
        const bool w4_small_sc = w4_sc16 && wq_bit == 4 && splits > 1 && kt_per <= 32 && wl == 1 && !half && stages == 2;
#define DMA_LAUNCH(WQ, O32, ST)                                                                                     \
    do { if constexpr (WQ == 8 && ST >= 3) { if (wl == 6) { hipLaunchKernelGGL((gemm_dma_kernel<WQ, O32, ST, 6>), g2, dim3(768), 0, s, x, w, scale, M, N, K, y, ldy, n_tiles, m_tiles, map_mode, kt_per, ws); break; } } \
         if (wl == 5) hipLaunchKernelGGL((gemm_dma_kernel<WQ, O32, ST, 5>), g2, dim3(512), 0, s, x, w, scale, M, N, K, y, ldy, n_tiles, m_tiles, map_mode, kt_per, ws); \
         else if (half) hipLaunchKernelGGL((gemm_dma_kernel<WQ, O32, ST, 1, 64>), g2, block, 0, s, x, w, scale, M, N, K, y, ldy, n_tiles, m_tiles, map_mode, kt_per, ws); \
         else if (WQ == 4 && ST == 2 && w4_small_sc) hipLaunchKernelGGL((gemm_dma_kernel<WQ, O32, ST, 1, G_BM, 16>), g2, block, 0, s, x, w, scale, M, N, K, y, ldy, n_tiles, m_tiles, map_mode, kt_per, ws); \
         else hipLaunchKernelGGL((gemm_dma_kernel<WQ, O32, ST, 1>), g2, block, 0, s, x, w, scale, M, N, K, y, ldy, n_tiles, m_tiles, map_mode, kt_per, ws); } while (0)
#define DMA_STAGES(WQ, O32)                                                                                         \
    do { if (stages == 2) DMA_LAUNCH(WQ, O32, 2); else if (stages == 3) DMA_LAUNCH(WQ, O32, 3); else DMA_LAUNCH(WQ, O32, 4); } while (0)
#define DMA_EPI(WQ) do { if (epi == EPI_F32) DMA_STAGES(WQ, EPI_F32); else if (epi == EPI_F16) DMA_STAGES(WQ, EPI_F16); else DMA_STAGES(WQ, EPI_SWIGLU); } while (0)
        if (wq_bit == 8) DMA_EPI(8); else if (wq_bit == 4) DMA_EPI(4); else DMA_EPI(0);
#undef DMA_EPI
#undef DMA_STAGES
#undef DMA_LAUNCH
        hipError_t e = hipGetLastError();
        if (e != hipSuccess || splits == 1) return e;
        if (defer && epi == EPI_F16 && N % 8 == 0 && splits <= 8) { *defer = SplitSlabs{ws, splits, wq_bit == 8 ? scale : nullptr, N, M}; return e; }
        return launch_splitk_reduce(s, ws, splits, M, N, wq_bit == 8 ? scale : nullptr, y, ldy, epi);
    }
#define GEMM_CASE(WQ, O32)                                                                                          \
    if (wq_bit == WQ && epi == (int)O32) {                                                                          \
        hipLaunchKernelGGL((gemm_kernel<WQ, O32>), grid, block, 0, s, x, w, scale, M, N, K, group, y, ldy, n_tiles, \
                           m_tiles);                                                                                \
        return hipGetLastError();                                                                                   \
    }
    GEMM_CASE(0, EPI_F16) GEMM_CASE(0, EPI_F32) GEMM_CASE(0, EPI_SWIGLU) GEMM_CASE(8, EPI_F16) GEMM_CASE(8, EPI_F32) GEMM_CASE(8, EPI_SWIGLU)
    GEMM_CASE(4, EPI_F16) GEMM_CASE(4, EPI_F32) GEMM_CASE(4, EPI_SWIGLU)
#undef GEMM_CASE
    return hipErrorInvalidValue;
}

}  // namespace pplhip
