// W8A16 tile GEMM for prefill-sized steps (M >= 3584), round 6: 256 x 256 x 64 block tile on FOUR waves, one per SIMD, each 64 (n) x 256 (m)
// on v_mfma_f32_32x32x16_f16 with its 32 accumulators (256 registers) resident.
//
// Why.  SQ counters of gemm_w8_dma256_kernel beside the vendor library's kernel on the same shapes (profiles/r06_gemm_bigm_counters.md): both
// issue 16-cycle MFMAs, the vendor's kernel 0.7 other instructions per MFMA, ours 2.25 (0.94 VALU: the int8 -> fp16 conversion and addresses;
// 0.54 LDS: eight waves each re-read the whole activation tile; 0.77 SALU: M0 save / restore around every LDS-DMA piece, loop and wait
// bookkeeping).  A SIMD issues about one instruction per 8.5 cycles from such a stream however many waves it holds (profiles/r05_valu_roles_
// probe.log, r06_gqa_experiments.md): 3.25 x 8.5 = 27.6 cycles per 16-cycle MFMA = the 55-58 % of the MFMA rate the kernel measures, and the
// vendor's 1.7 x 8.5 = 14.6 is why it is MFMA-bound.  So: instructions per flop.  Here a wave's tile is 64 x 256 per 16-deep k-step -- 2 weight
// + 8 activation fragment reads and 2 conversions for 16 MFMAs of 32 cycles -- every int8 weight is converted ONCE per block, the LDS-DMA
// pieces take a scalar base + 32-bit lane offset and leave M0 clobbered (2-3 instructions per KiB instead of 5), fragment addresses are four
// registers per operand plus immediates.  Per K tile and wave: 64 MFMAs and ~180 other instructions = 2.8 per 32-cycle MFMA (budget 2.76).
// LDS image, swizzles and fragment addressing as in k_gemm_asm (probes) / k_gemm_pc: activations [256 rows][64 fp16], 16-byte chunk q of row
// r at position q ^ ((r >> 1) & 7); weights [256 rows][64 int8], chunk c of row r at position c ^ ((r >> 2) & 3).  MFMA k-step s: lane (r = l &
// 31, h = l >> 5) multiplies k = 16 s + 8 h .. + 8; accumulator value e of lane (r, h) = channel 8 (e >> 2) + 4 h + (e & 3) of the 32-channel
// block, row r of the 32-row block.
// Oracle: ref_linear_raw (oracle/llama_ref.c).
#include <stdlib.h>
#include "k_gemm_dev.h"

namespace pplhip {

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));
constexpr int BG_BM = 256, BG_BN = 256, BG_ST = 3;
constexpr int BG_XB = BG_BM * G_BK * 2, BG_WB = BG_BN * G_BK;   // bytes per stage: 32 KiB + 16 KiB
constexpr int BG_LDS = BG_ST * (BG_XB + BG_WB);                 // 144 KiB

// one KiB of a tile row group, global -> LDS: scalar base + unsigned 32-bit lane offset; M0 is left clobbered (nothing else in this kernel
// uses it: LDS instructions have not needed M0 since gfx9)
__device__ __forceinline__ void glds16_s(uint32_t voff, const void* sbase, uint32_t lds_wave_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_wave_base) : "memory");
}

// NW = 4: one wave per SIMD, 64 (n) x 256 (m) per wave; NW = 8: two per SIMD, 32 (n) x 256 (m) per wave (the wave layout of gemm_w8_dma256_kernel on
// the larger MFMA: half its MFMA instructions, the same fragment reads and conversions)
template <int EPI, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_w8_big_kernel(const uint16_t* __restrict__ x, const int8_t* __restrict__ w,
                                                          const uint16_t* __restrict__ scale, int64_t M, int N, int K, void* __restrict__ yv,
                                                          int64_t ldy, int n_tiles, int m_tiles, int gn, int gm) {
    extern __shared__ __attribute__((aligned(128))) char smem_bg[];   // ST x activations, then ST x weights; reused to stage the output
    // block -> tile: XCD id % 8 owns the weight tiles n == xcd (mod 8) and walks them in super-tiles of gn (n) x gm (m), m fastest
    // (k_gemm.hip gemm_w8_dma256_kernel: the ~32 blocks an XCD runs at a time share operand tiles in its L2)
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int per_super = gn * gm, sm_count = m_tiles / gm;
    const int sup = slot / per_super, within = slot % per_super;
    const int nt = xcd + 8 * ((sup / sm_count) * gn + within / gm);
    const int mt = (sup % sm_count) * gm + within % gm;
    if (nt >= n_tiles) return;
    const int n0 = nt * BG_BN;
    const int64_t m0 = (int64_t)mt * BG_BM;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ktiles = K / G_BK;

    constexpr int NI = 8 / NW;            // 32-channel weight fragments per wave and k-step
    constexpr int XPW = 32 / NW, WPW = 16 / NW;   // LDS-DMA pieces per wave and stage
    // ---- LDS-DMA: 48 one-KiB pieces per stage, wave w issues activation pieces w + NW j (rows 8 P .. + 8) and weight pieces w + NW j (rows 16 P .. + 16)
    uint32_t xoff[XPW], woff[WPW];
#pragma unroll
    for (int j = 0; j < XPW; ++j) {
        const int P = wave + NW * j;
        const int p = P * 64 + lane, row = p >> 3, c = (p & 7) ^ ((row >> 1) & 7);
        int64_t m = m0 + row;
        if (m >= M) m = M - 1;
        xoff[j] = (uint32_t)((m * K + c * 8) * 2);
    }
#pragma unroll
    for (int j = 0; j < WPW; ++j) {
        const int Pw = wave + NW * j;
        const int p = Pw * 64 + lane, row = p >> 2, c = (p & 3) ^ ((row >> 2) & 3);
        int n = n0 + row;
        if (n >= N) n = N - 1;
        woff[j] = (uint32_t)((int64_t)n * K + c * 16);
    }
    const uint32_t xl = __builtin_amdgcn_readfirstlane(lds_addr(smem_bg) + wave * 1024);
    const uint32_t wl = __builtin_amdgcn_readfirstlane(lds_addr(smem_bg) + BG_ST * BG_XB + wave * 1024);
    auto issue = [&](int stage, int kt) {
        const char* xb = reinterpret_cast<const char*>(x) + (int64_t)kt * (G_BK * 2);
        const char* wb = reinterpret_cast<const char*>(w) + (int64_t)kt * G_BK;
#pragma unroll
        for (int j = 0; j < XPW; ++j) glds16_s(xoff[j], xb, xl + stage * BG_XB + j * (NW * 1024));
#pragma unroll
        for (int j = 0; j < WPW; ++j) glds16_s(woff[j], wb, wl + stage * BG_WB + j * (NW * 1024));
    };

    // ---- fragment addresses: one register per k-step and operand, everything else is an immediate
    const int r = lane & 31, h = lane >> 5;
    const char* xa[4];
    const char* wa_[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        xa[s] = smem_bg + r * 128 + (((2 * s + h) ^ ((r >> 1) & 7)) << 4);
        const int row = wave * (32 * NI) + r;                            // (row + 32: the same swizzle -- (row >> 2) & 3 only sees bits 2, 3)
        wa_[s] = smem_bg + BG_ST * BG_XB + row * 64 + ((s ^ ((row >> 2) & 3)) << 4) + h * 8;
    }
    auto ldsx = [&](const char* a) { return *reinterpret_cast<const uint4*>(a); };
    auto ldsw = [&](const char* a) { return *reinterpret_cast<const uint2*>(a); };

    f16v acc[NI][8];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    issue(0, 0);
    if (ktiles > 1) issue(1, 1);
    int st = 0, stn = 2;
    for (int t = 0; t < ktiles; ++t) {
        if (t + 1 < ktiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XPW + WPW) : "memory");   // tile t landed (tile t + 1's pieces may be in flight)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // tile t is published; the stage read during iteration t - 1 (= stage of tile t + 2) is free
        if (t + 2 < ktiles) issue(stn, t + 2);
        const uint32_t xo = st * BG_XB, wo = st * BG_WB;
        st = st == BG_ST - 1 ? 0 : st + 1;
        stn = stn == BG_ST - 1 ? 0 : stn + 1;
        // Two fragment sets: the reads of k-step s + 1 are issued between the MFMAs of k-step s (one read behind every two MFMAs), its weights
        // are converted behind the last MFMA.  Pinned with sched_group_barrier: left alone hipcc keeps ONE activation fragment register set and
        // waits out every LDS read in front of the two MFMAs that use it (a lone wave per SIMD has nobody to fill those gaps).
        h8 xb[2][8];
        uint2 wr[2][NI];
        h8 wf[NI];
#pragma unroll
        for (int j = 0; j < 8; ++j) xb[0][j] = __builtin_bit_cast(h8, ldsx(xa[0] + xo + j * 4096));
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            wr[0][i] = ldsw(wa_[0] + wo + i * (32 * 64));
            wf[i] = cvt_i8x8_f16(wr[0][i]);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            if (s < 3) {
#pragma unroll
                for (int i = 0; i < NI; ++i) wr[nxt][i] = ldsw(wa_[s + 1] + wo + i * (32 * 64));
#pragma unroll
                for (int j = 0; j < 8; ++j) xb[nxt][j] = __builtin_bit_cast(h8, ldsx(xa[s + 1] + xo + j * 4096));
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int i = 0; i < NI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i], xb[cur][j], acc[i][j], 0, 0, 0);
            if (s < 3) {
                __builtin_amdgcn_sched_group_barrier(0x100, NI, 0);      // the raw weight reads first
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    __builtin_amdgcn_sched_group_barrier(0x008, NI, 0);  // NI MFMAs
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one activation fragment read
                }
#pragma unroll
                for (int i = 0; i < NI; ++i) wf[i] = cvt_i8x8_f16(wr[nxt][i]);
            }
        }
    }

    // ---- epilogue through LDS (the ring is idle): whole 16-byte row pieces leave the block (k_gemm.hip gemm_w8_dma256_kernel)
    constexpr int OUTW = EPI == EPI_SWIGLU ? BG_BN / 2 : BG_BN, ROWB = OUTW * 2 + 16, CPR = OUTW / 8;
    static_assert(BG_BM * ROWB <= BG_LDS, "the staging image fits the ring");
    char* const stg = smem_bg;
    __syncthreads();   // every wave is past its last fragment read
#pragma unroll
    for (int i = 0; i < NI; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int nl = wave * (32 * NI) + 32 * i + 8 * q + 4 * h;
            const int nc = n0 + nl < N ? n0 + nl : 0;
            const h4 sh = __builtin_bit_cast(h4, *reinterpret_cast<const uint2*>(scale + nc));
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v0 = acc[i][j][4 * q] * (float)sh[0], v1 = acc[i][j][4 * q + 1] * (float)sh[1], v2 = acc[i][j][4 * q + 2] * (float)sh[2],
                            v3 = acc[i][j][4 * q + 3] * (float)sh[3];
                char* dst = stg + (32 * j + r) * ROWB;
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
    }
    __syncthreads();
    uint16_t* const y = reinterpret_cast<uint16_t*>(yv);
    const int nout = EPI == EPI_SWIGLU ? N / 2 : N, c0 = EPI == EPI_SWIGLU ? n0 / 2 : n0;
#pragma unroll 4
    for (int c = tid; c < BG_BM * CPR; c += NW * 64) {
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
}

}  // namespace

// W8A16, K % 64 == 0, fp16 or fused-SwiGLU output with 16-byte-aligned rows; x / w / scale 16- / 16- / 8-byte aligned; M K 2 and N K below 4 GiB
bool linear_w8_big_supported(int64_t M, int N, int K, const void* x, const void* w, const void* scale, const void* y, int64_t ldy, int epi) {
    if (K % G_BK || N % 4 || (epi != EPI_F16 && epi != EPI_SWIGLU)) return false;
    if (ldy % 8 || ((uintptr_t)y & 15) || ((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)scale & 7)) return false;
    if (epi == EPI_SWIGLU && N % 16) return false;
    if ((uint64_t)M * (uint64_t)K * 2 >= (1ull << 32) || (uint64_t)N * (uint64_t)K >= (1ull << 32)) return false;
    return true;
}

hipError_t launch_linear_w8_big(hipStream_t s, const uint16_t* x, const int8_t* w, const uint16_t* scale, int64_t M, int N, int K, void* y,
                                int64_t ldy, int epi) {
    const int nt = (N + BG_BN - 1) / BG_BN, mt = (int)((M + BG_BM - 1) / BG_BM);
    static bool attr_dev[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_dev[dev & 63]) {
        (void)hipFuncSetAttribute((const void*)gemm_w8_big_kernel<EPI_F16, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, BG_LDS);
        (void)hipFuncSetAttribute((const void*)gemm_w8_big_kernel<EPI_SWIGLU, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, BG_LDS);
        (void)hipFuncSetAttribute((const void*)gemm_w8_big_kernel<EPI_F16, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, BG_LDS);
        (void)hipFuncSetAttribute((const void*)gemm_w8_big_kernel<EPI_SWIGLU, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, BG_LDS);
        attr_dev[dev & 63] = true;
    }
    // super-tiles as for gemm_w8_dma256_kernel: gm = the largest divisor of mt <= 4, gn = up to 8 weight tiles of the XCD's share
    const int nl = (nt + 7) / 8;
    int gm = 4;
    while (gm > 1 && mt % gm) --gm;
    int gn = 8;
    if (gn > nl) gn = nl;
    const int nl_pad = (nl + gn - 1) / gn * gn;
    dim3 grid((unsigned)(8 * nl_pad * mt));
    static const int nw = getenv("PPLHIP_GEMM_BIG") ? atoi(getenv("PPLHIP_GEMM_BIG")) : 8;   // waves per block: 8 (default) or 4
#define BG_L(E, W) hipLaunchKernelGGL((gemm_w8_big_kernel<E, W>), grid, dim3(W * 64), BG_LDS, s, x, w, scale, M, N, K, y, ldy, nt, mt, gn, gm)
    if (nw == 4) { if (epi == EPI_SWIGLU) BG_L(EPI_SWIGLU, 4); else BG_L(EPI_F16, 4); }
    else { if (epi == EPI_SWIGLU) BG_L(EPI_SWIGLU, 8); else BG_L(EPI_F16, 8); }
#undef BG_L
    return hipGetLastError();
}

}  // namespace pplhip
