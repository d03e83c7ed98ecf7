// PROBE, not part of libpplhip.so (round 2): a 256(n) x 128(m) producer / consumer W8A16 tile kernel on v_mfma_f32_32x32x16_f16.
// Built into the library for the measurements in profiles/r02_gemm_experiments.md and removed again: per tile it is faster than
// the 128 x 128 kernel (51 % of the dense fp16 peak against ~40 %), but at M = 1024 its 384 / 688 tiles run as 2 / 3 rounds of the
// 256 one-block-per-CU slots (75 % / 90 % filled), which leaves wqkv at 129 us (128 x 128 kernel: 104 us) and w13 at 214 us (203 us).
// To rebuild: add it to csrc/Makefile, declare launch_gemm_w8_pc in kernels.h and call it from launch_linear for wq_bit == 8,
// M >= 512, N >= 8192.
// The producer / consumer W8A16 tile kernel for the decode step of a full running batch (see below).  A translation unit of
// its own: k_gemm.hip is built with -amdgpu-mfma-vgpr-form=1 (accumulators in architectural VGPRs), this kernel keeps the
// compiler's default accumulator placement.
#include "k_gemm_dev.h"

namespace pplhip {

// ---------------------------------------------------------------------------------------------------------------
// W8A16, 512 <= M < 4096 (the decode step of a full running batch, M = 1024) with N >= 8192 (wqkv, w13):
// 256(n) x 128(m) x 64(k) block tile, one block of 8 waves per CU: 4 CONSUMER waves stacked along n, each a
// 64(n) x 128(m) wave tile on v_mfma_f32_32x32x16_f16, and 4 PRODUCER waves that only issue the LDS-DMA of a 4-stage ring.
//   * what the matrix pipe of the 128 x 128 kernel waits for (profiles/r02_gemm_*): the LDS -- its 32(n) x 128(m) wave
//     tiles read 288 B of fragments per k, 2304 B per k and CU = 112 % of the matrix time -- and the issue of the LDS-DMA
//     pieces (100 - 185 cycles each for the issuing wave, MI355X_MICROARCH.md), in front of the same wave's MFMAs.
//     Here a consumer wave is twice as tall in n, where a row costs ONE byte per k (int8 weights stay int8 in LDS): 320 B
//     per k and wave for twice the flops (1280 B per k and CU, 62 %), every activation fragment feeds two matrix
//     instructions, and no consumer ever issues a load from memory;
//   * the 32x32x16 MFMA is the form that reaches the dense fp16 peak on gfx950 (microbench 2.2-2.5 PFLOP/s against 1.96 for
//     16x16x32) and needs half as many issue slots per flop: a consumer's K tile is 32 matrix instructions, 20 fragment
//     reads and 8 fragment conversions (int8 -> fp16, exact, 10 VALU each) = 3.4 other instructions per MFMA;
//   * k order inside a 64-deep tile: lane half h of the MFMA (lanes 32 h .. 32 h + 31) multiplies k = 32 h + 8 s .. + 8 in
//     matrix step s -- for both operands, so the products pair up exactly -- which makes a lane's int8 operands of all
//     four steps 32 CONTIGUOUS bytes of its weight row (two ds_read_b128 per 32 rows and tile) and its activation operand
//     of step s the 16-byte chunk 4 h + s of its activation row; activation fragments are fetched one step ahead.
//   * one barrier per K tile: it publishes tile t (every producer has waited for its own pieces) and frees the stage of
//     tile t - 1 for the DMA of tile t + 3.
// Numerics: identical to the other tile kernels (exact int8 -> fp16, fp32 accumulate, scale and one rounding in the
// epilogue).
// ---------------------------------------------------------------------------------------------------------------
typedef float f16v __attribute__((ext_vector_type(16)));
#ifndef PC_ABLATE
#define PC_ABLATE 0   // diagnosis builds only (-DPC_ABLATE=bits: 1 no LDS-DMA, 2 no activation fragment reads, 4 no int8 conversion)
#endif
constexpr int P_BN = 256, P_BM = 128;
constexpr int P_SUB = 2;   // 64-deep sub-tiles per ring stage = per barrier (a barrier costs the consumers an LDS round trip + a conversion)
constexpr int P_ST = 2;    // ring stages
constexpr int P_W_STAGE = P_BN * G_BK, P_X_STAGE = P_BM * G_BK * 2, P_STAGE = P_X_STAGE + P_W_STAGE;  // one sub-tile: 16 + 16 KiB
constexpr int P_LDS = P_ST * P_SUB * P_STAGE;

template <int EPI>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_w8_pc_kernel(const uint16_t* __restrict__ x, const int8_t* __restrict__ w,
                                                         const uint16_t* __restrict__ scale, int64_t M, int N, int K,
                                                         void* __restrict__ yv, int64_t ldy, int n_tiles, int m_tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem_p[];  // P_ST x (X 16 KiB | W 16 KiB)
    constexpr int NI = 2, NJ = 4;   // consumer wave tile: 2 x 32 weight rows, 4 x 32 activation rows

    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int nt = xcd + 8 * (slot / m_tiles);
    const int mt = slot % m_tiles;
    if (nt >= n_tiles) return;
    const int n0 = nt * P_BN;
    const int64_t m0 = (int64_t)mt * P_BM;
    const int ktiles = K / G_BK;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    if (wave >= 4) {
        // ---- producers: 4 waves x (4 activation + 4 weight pieces of 1 KiB) per tile --------------------------------
        const int pt = threadIdx.x - 256;  // 0 .. 255
        const uint16_t* xsrc[4];
        const int8_t* wsrc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            {   // LDS is written linearly; the chunk swizzle sits on the source address and on the fragment reads
                const int p = j * 256 + pt, row = p >> 3, c = (p & 7) ^ ((row >> 1) & 7);  // LDS position p & 7 holds k chunk c
                int64_t m = m0 + row;
                if (m >= M) m = M - 1;
                xsrc[j] = x + m * K + c * 8;
            }
            {
                const int p = j * 256 + pt, row = p >> 2, c = (p & 3) ^ ((row >> 2) & 3);
                int n = n0 + row;
                if (n >= N) n = N - 1;
                wsrc[j] = w + (int64_t)n * K + c * 16;
            }
        }
        const uint32_t xdst = __builtin_amdgcn_readfirstlane(lds_addr(smem_p) + (wave - 4) * 1024);
        const uint32_t wdst = xdst + P_X_STAGE;
        auto issue = [&](int stage, int k0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) glds16(xsrc[j] + k0, xdst + stage * P_STAGE + j * 4096);
#pragma unroll
            for (int j = 0; j < 4; ++j) glds16(wsrc[j] + k0, wdst + stage * P_STAGE + j * 4096);
        };
        // stage g (P_SUB sub-tiles) is issued right after the barrier that frees its buffer (the one that publishes stage
        // g - 1) and waited for in front of the next barrier: it has one whole stage of matrix work to land
        const int nst = (ktiles + P_SUB - 1) / P_SUB;
        auto issue_stage = [&](int g) {
#pragma unroll
            for (int u = 0; u < P_SUB; ++u)
                if (g * P_SUB + u < ktiles && !(PC_ABLATE & 1)) issue((g & 1) * P_SUB + u, (g * P_SUB + u) * G_BK);
        };
        issue_stage(0);
        for (int g = 0; g < nst; ++g) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (g + 1 < nst) issue_stage(g + 1);
        }
        return;
    }

    // ---- consumers ---------------------------------------------------------------------------------------------------
    const int l31 = lane & 31, hh = lane >> 5;
    const int nb = wave * 64;       // first weight row of this wave inside the tile
    f16v acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // byte offsets of this lane's fragments inside a stage (the swizzles depend on the row only)
    int woff[NI][2], xoff[NJ][4];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int row = nb + i * 32 + l31;
#pragma unroll
        for (int q = 0; q < 2; ++q) woff[i][q] = P_X_STAGE + row * G_BK + (((hh * 2 + q) ^ ((row >> 2) & 3)) * 16);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int row = j * 32 + l31;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) xoff[j][s4] = row * (G_BK * 2) + (((hh * 4 + s4) ^ ((row >> 1) & 7)) * 16);
    }
    for (int t = 0; t < ktiles; ++t) {
        // one barrier per stage: it publishes the stage's sub-tiles (the producers waited for their pieces) and tells the
        // producers that the other stage is no longer read
        if (t % P_SUB == 0) __syncthreads();
        const char* sb = smem_p + (t % (P_ST * P_SUB)) * P_STAGE;
        // Software pipeline inside the tile: all 4 weight reads + the activation fragments of step 0, convert step 0; step s:
        // the 4 activation reads and the conversions of step s + 1 are issued in front of the 8 MFMAs of step s.
        // (amdgpu_waves_per_eu(2, 2) on the kernel matters: aiming for a third wave per SIMD hipcc otherwise funnels every
        // activation fragment through ONE register quad -- read, wait for the LDS, two MFMAs, read the next -- and the
        // matrix pipe idles for an LDS round trip after every second instruction: 53 % duty measured.)
        uint4 wraw[NI][2];
        h8 b[2][NJ], a[2][NI];
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int q = 0; q < 2; ++q) wraw[i][q] = *reinterpret_cast<const uint4*>(sb + woff[i][q]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) b[0][j] = (PC_ABLATE & 2) ? h8{} : __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(sb + xoff[j][0]));
        auto cvt = [&](int i, int s4) {
            const uint4 v = wraw[i][s4 >> 1];
            return (PC_ABLATE & 4) ? __builtin_bit_cast(h8, v) : cvt_i8x8_f16((s4 & 1) ? make_uint2(v.z, v.w) : make_uint2(v.x, v.y));
        };
#pragma unroll
        for (int i = 0; i < NI; ++i) a[0][i] = cvt(i, 0);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            if (s4 < 3) {
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    b[(s4 + 1) & 1][j] = (PC_ABLATE & 2) ? b[s4 & 1][j] : __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(sb + xoff[j][s4 + 1]));
#pragma unroll
                for (int i = 0; i < NI; ++i) a[(s4 + 1) & 1][i] = cvt(i, s4 + 1);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int i = 0; i < NI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s4 & 1][i], b[s4 & 1][j], acc[i][j], 0, 0, 0);
        }
    }

    // C layout of the 32 x 32 MFMA: lane (l31, hh) holds, for activation row m = .. + l31, weight rows n = .. + 8 q + 4 hh + r
    // (q = reg >> 2, r = reg & 3): four consecutive channels per q -> 8-byte stores and 8-byte scale loads
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = n0 + nb + i * 32 + q * 8 + hh * 4;
            if (n >= N) continue;
            const h4 sh = __builtin_bit_cast(h4, *reinterpret_cast<const uint2*>(scale + n));
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int64_t m = m0 + j * 32 + l31;
                if (m >= M) continue;
                store4<EPI>(yv, ldy, m, n, acc[i][j][q * 4 + 0] * (float)sh[0], acc[i][j][q * 4 + 1] * (float)sh[1],
                            acc[i][j][q * 4 + 2] * (float)sh[2], acc[i][j][q * 4 + 3] * (float)sh[3]);
            }
        }
}


hipError_t launch_gemm_w8_pc(hipStream_t s, const uint16_t* x, const int8_t* w, const uint16_t* scale, int64_t M, int N, int K, void* y,
                             int64_t ldy, int epi) {
    const int nt2 = (N + P_BN - 1) / P_BN, mt2 = (int)((M + P_BM - 1) / P_BM);
    dim3 gp((unsigned)((nt2 + 7) / 8 * 8 * mt2));
    static bool attr_dev_p[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_dev_p[dev & 63]) {
        (void)hipFuncSetAttribute((const void*)gemm_w8_pc_kernel<EPI_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS);
        (void)hipFuncSetAttribute((const void*)gemm_w8_pc_kernel<EPI_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS);
        (void)hipFuncSetAttribute((const void*)gemm_w8_pc_kernel<EPI_SWIGLU>, hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS);
        attr_dev_p[dev & 63] = true;
    }
#define PL(E) hipLaunchKernelGGL((gemm_w8_pc_kernel<E>), gp, dim3(512), P_LDS, s, x, w, scale, M, N, K, y, ldy, nt2, mt2)
    if (epi == EPI_F32) PL(EPI_F32); else if (epi == EPI_F16) PL(EPI_F16); else PL(EPI_SWIGLU);
#undef PL
    return hipGetLastError();
}

}  // namespace pplhip
