// W8A16 tile GEMM with a 128 (m) x 384 (n) x 64 block tile: 12 waves, each 32 (n) x 128 (m) like the waves of gemm_dma_kernel, but the
// THREE wave quads that used to be three independent 128 x 128 blocks of a CU now share one activation tile in LDS.
//
// Why (profiles/r03_gemm_experiments.md #16): the 128 x 128 kernel moves 24 KiB into LDS per 512 MFMA cycles and CU -- 47 B / clk of the
// 64 B / clk a CU's vector-memory path delivers -- and an ablation shows its waves never wait for those bytes to ARRIVE (issuing the
// LDS-DMA and never waiting for it changes nothing) while not issuing it at all makes the kernel 30 % faster: the tile traffic itself
// is the co-bottleneck.  One 128 x 384 block moves 40 KiB per 1536 MFMA cycles: 26 B / clk.
// Tile counts at M = 1024: wqkv (N = 12288) 32 x 8 = 256 blocks = one per CU; w13 (N = 22016) 58 x 8 = 464.
#include <stdlib.h>
#include "k_gemm_dev.h"

namespace pplhip {

namespace {

constexpr int WD_BM = 128, WD_NP = 4;
constexpr int WD_XB = WD_BM * G_BK * 2;   // activation bytes per stage: 16 KiB
constexpr int WD_XP = WD_XB / 1024;       // = 16 one-KiB DMA pieces

// 16 waves: 12 consumers (wave w: weight rows 32 w .. + 32 x all 128 activation rows) and 4 producers (one per SIMD) that do nothing
// but keep the ST-stage ring filled.
// NC consumer waves: block tile 128 (m) x 32 NC (n); NC = 12, 11, 10 so that the tile count fits whole rounds of 256 blocks
template <int EPI, int ST, int NC, int ABL = 0>
__global__ __launch_bounds__((NC + WD_NP) * 64) void gemm_w8_wide_kernel(const uint16_t* __restrict__ x, const int8_t* __restrict__ w,
                                                                  const uint16_t* __restrict__ scale, int64_t M, int N, int K,
                                                                  void* __restrict__ yv, int64_t ldy, int n_tiles, int m_tiles) {
    constexpr int WD_BN = 32 * NC, WD_NC = NC;
    constexpr int WD_WB = WD_BN * G_BK, WD_WP = WD_WB / 1024;           // weight bytes per stage (24 KiB at NC = 12) = 2 NC pieces
    constexpr int WD_PP = (WD_XP + WD_WP + WD_NP - 1) / WD_NP;          // pieces per producer wave and tile: 10 (the last ones may be missing)
    extern __shared__ __attribute__((aligned(16))) char smem_wd[];  // ST x (X 16 KiB) then ST x (W 2 NC KiB)
    char* const Xs0 = smem_wd;
    char* const Wq0 = smem_wd + ST * WD_XB;

    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;  // XCD id % 8 owns the weight tiles n == xcd (mod 8) and all their m tiles
    const int nt = xcd + 8 * (slot / m_tiles);
    const int mt = slot % m_tiles;
    if (nt >= n_tiles) return;
    const int n0 = nt * WD_BN;
    const int64_t m0 = (int64_t)mt * WD_BM;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ktiles = K / G_BK;
    constexpr int D = ST - 1;  // prefetch distance

    if (wave >= WD_NC) {
        // ---- producer p: pieces P = p + 4 j of every tile; j < 4: activation piece P (rows 8 P .. + 8), j >= 4: weight piece P - 16
        // (rows 16 (P - 16) .. + 16).  Source swizzles as in gemm_dma_body (the DMA writes LDS linearly).
        const int pw = wave - WD_NC;
        const char* psrc[WD_PP];
        uint32_t pdst[WD_PP];
        const uint32_t xbase = lds_addr(Xs0), wbase = lds_addr(Wq0);
        const int npc = (WD_XP + WD_WP - pw + WD_NP - 1) / WD_NP;  // this producer's pieces per tile: WD_PP or WD_PP - 1
#pragma unroll
        for (int j = 0; j < WD_PP; ++j) {
            const int P = pw + WD_NP * j;
            if (j < WD_XP / WD_NP) {
                const int p = P * 64 + lane, row = p >> 3, pos = (p & 7) ^ ((row >> 1) & 7);
                const int c = ((pos & 3) << 1) | (pos >> 2);
                int64_t m = m0 + row;
                if (m >= M) m = M - 1;
                psrc[j] = reinterpret_cast<const char*>(x + m * K + c * 8);
                pdst[j] = __builtin_amdgcn_readfirstlane(xbase + P * 1024);
            } else {
                const int Pw = (P < WD_XP + WD_WP ? P : WD_XP) - WD_XP;  // (a piece past the tile is never issued)
                const int p = Pw * 64 + lane, row = p >> 2, c = (p & 3) ^ w_swz(row);
                int n = n0 + row;
                if (n >= N) n = N - 1;
                psrc[j] = reinterpret_cast<const char*>(w) + (int64_t)n * K + c * 16;
                pdst[j] = __builtin_amdgcn_readfirstlane(wbase + Pw * 1024);
            }
        }
#define WD_PRODUCE(KT, STG)                                                                                                          \
    do {                                                                                                                             \
        _Pragma("unroll") for (int j = 0; j < WD_PP; ++j)                                                                            \
            if (j < npc)                                                                                                             \
            glds16(psrc[j] + (int64_t)(KT) * (j < WD_XP / WD_NP ? G_BK * 2 : G_BK), pdst[j] + (STG) * (j < WD_XP / WD_NP ? WD_XB : WD_WB)); \
    } while (0)
        // ST-stage ring, prefetch distance D = ST - 1: while the consumers multiply tile t, tiles t + 1 .. t + D are in flight
        // (measured against a barrier in the middle of the consumers' iteration with distance 1: 94-97 vs 98-99 us for wqkv)
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (d < ktiles) WD_PRODUCE(d, d);
        int stn = D % ST;
        for (int t = 0; t < ktiles; ++t) {
            const int younger = (ktiles - 1 - t) < (D - 1) ? (ktiles - 1 - t) : (D - 1);
            if (younger >= 2) {
                if (npc == WD_PP) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * WD_PP) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * WD_PP - 2) : "memory");
            } else if (younger == 1) {
                if (npc == WD_PP) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WD_PP) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WD_PP - 1) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (!(ABL & 8)) __syncthreads();  // tile t is published; the stage the consumers read during iteration t - 1 is free
            if (t + D < ktiles && !(ABL & 1)) WD_PRODUCE(t + D, stn);
            stn = stn == ST - 1 ? 0 : stn + 1;
        }
#undef WD_PRODUCE
        return;
    }

    // ---- consumers ----------------------------------------------------------------------------------------------------------------
    const int l15 = lane & 15, kq = lane >> 4;
    const int nb = wave * 32;
    f4 acc[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    int st = 0;
    for (int t = 0; t < ktiles; ++t) {
        if (!(ABL & 8)) __syncthreads();  // tile t is published
        const uint16_t* xs = reinterpret_cast<const uint16_t*>(Xs0 + st * WD_XB);
        const char* wq = Wq0 + st * WD_WB;
        st = st == ST - 1 ? 0 : st + 1;
        uint4 wraw[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = nb + i * 16 + l15;
            wraw[i] = *reinterpret_cast<const uint4*>(&wq[row * G_BK + (kq ^ w_swz(row)) * 16]);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            h8 a[2], bfr[8];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if constexpr (ABL & 2) a[i] = __builtin_bit_cast(h8, make_uint4(wraw[i].x, wraw[i].y, wraw[i].z ^ ks, wraw[i].w));
                else a[i] = cvt_i8x8_f16(ks == 0 ? make_uint2(wraw[i].x, wraw[i].y) : make_uint2(wraw[i].z, wraw[i].w));
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = j * 16 + l15;
                bfr[j] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(&xs[((ABL & 4) ? 0 : row * G_BK) + g_swz(row, ks * 4 + kq) * 8]));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }

#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int n = n0 + nb + i * 16 + kq * 4;
        if (n >= N) continue;
        const h4 sh = __builtin_bit_cast(h4, *reinterpret_cast<const uint2*>(scale + n));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t m = m0 + j * 16 + l15;
            if (m >= M) continue;
            store4<EPI>(yv, ldy, m, n, acc[i][j][0] * (float)sh[0], acc[i][j][1] * (float)sh[1], acc[i][j][2] * (float)sh[2],
                        acc[i][j][3] * (float)sh[3]);
        }
    }
}

}  // namespace

// 12 consumer waves when the 128 x 384 tiles fill whole rounds of 256 one-per-CU blocks well enough, else 0 (the 128 x 128 kernel with its three
// independent blocks per CU is the better choice).  Narrower blocks were measured and lose: w13 at M = 1024 (N = 22016) with 11 consumer waves
// (504 blocks, two almost full rounds) 205-208 us against 191-194 us with 12 (464 blocks) -- the round time does not shrink with the tile.
int linear_w8_wide_waves(int64_t M, int N) {
    const int64_t m_tiles = (M + WD_BM - 1) / WD_BM;
    const int64_t tiles = (N + 383) / 384 * m_tiles, rounds = (tiles + 255) / 256;
    const double eff = (double)N * (double)M / ((double)rounds * 256.0 * 384.0 * WD_BM);
    // measured: with more than two rounds the 128 x 128 kernel catches up (w13 at M = 2048: 4 rounds 382 us against 367 us) unless the fit is exact
    return (eff >= 0.85 && (rounds <= 2 || eff >= 0.97)) ? 12 : 0;
}

// W8A16, K % 64 == 0, N % 4 == 0.  epi: EPI_F16 / EPI_F32 / EPI_SWIGLU; nc from linear_w8_wide_waves
hipError_t launch_linear_w8_wide(hipStream_t s, const uint16_t* x, const int8_t* w, const uint16_t* scale, int64_t M, int N, int K, void* y,
                                 int64_t ldy, int epi, int nc) {
    if (nc != 12) return hipErrorInvalidValue;
    const int bn = 32 * nc;
    const int n_tiles = (N + bn - 1) / bn, m_tiles = (int)((M + WD_BM - 1) / WD_BM);
    constexpr int ST = 3;
    const size_t lds = (size_t)ST * (WD_XB + bn * G_BK);
    static bool attr_dev[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_dev[dev & 63]) {
#define WD_A(E, C) (void)hipFuncSetAttribute((const void*)gemm_w8_wide_kernel<E, ST, C>, hipFuncAttributeMaxDynamicSharedMemorySize, ST * (WD_XB + 32 * C * G_BK))
        WD_A(EPI_F16, 12); WD_A(EPI_F32, 12); WD_A(EPI_SWIGLU, 12);
#undef WD_A
        attr_dev[dev & 63] = true;
    }
    dim3 grid((unsigned)((n_tiles + 7) / 8 * 8 * m_tiles));
#ifdef WD_ABLATE_BUILD  // diagnosis (wrong results): PPLHIP_GEMM_WIDE_ABLATE = 1 no refills, 2 no conversion, 4 one activation row, 8 no barriers
    static const int abl = getenv("PPLHIP_GEMM_WIDE_ABLATE") ? atoi(getenv("PPLHIP_GEMM_WIDE_ABLATE")) : 0;   // (an ablation build reads its switch itself: no TUNING=1 needed)
#define WD_AB(A) if (abl == A) { (void)hipFuncSetAttribute((const void*)gemm_w8_wide_kernel<EPI_F16, ST, 12, A>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((gemm_w8_wide_kernel<EPI_F16, ST, 12, A>), grid, dim3((12 + WD_NP) * 64), lds, s, x, w, scale, M, N, K, y, ldy, n_tiles, m_tiles); return hipGetLastError(); }
    WD_AB(1) WD_AB(2) WD_AB(3) WD_AB(4) WD_AB(6) WD_AB(7) WD_AB(8) WD_AB(15)
#undef WD_AB
#endif
#define WD_L(E, C) hipLaunchKernelGGL((gemm_w8_wide_kernel<E, ST, C>), grid, dim3((C + WD_NP) * 64), lds, s, x, w, scale, M, N, K, y, ldy, n_tiles, m_tiles)
#define WD_E(C) do { if (epi == EPI_F32) WD_L(EPI_F32, C); else if (epi == EPI_F16) WD_L(EPI_F16, C); else WD_L(EPI_SWIGLU, C); } while (0)
    WD_E(12);
#undef WD_E
#undef WD_L
    return hipGetLastError();
}

}  // namespace pplhip
