// PROBE, not part of libpplhip.so (round 2): built into the library for the measurement in profiles/r02_gemm_experiments.md and removed again.
// To rebuild: copy to csrc/k_gemm_8ph.hip, add k_gemm_8ph.o to csrc/Makefile, declare launch_gemm_w8_8ph in kernels.h, call it from launch_linear.
// W8A16 GEMM for large M (steps that carry prefill): 256(n) x 256(m) x 64(k) tiles on the "8-phase" schedule of
// cdna_hip_programming.md (256^2 template): 8 waves as 4 (n) x 2 (m), each a 64(n) x 128(m) wave tile = 4 quadrants of
// 16 x mfma_f32_16x16x32_f16 per K tile; one PHASE per quadrant = { fragment reads + conversions + LDS-DMA issue | barrier |
// MFMA cluster under s_setprio(1) | barrier }, and the two wave rows (wm = 0 / 1) run ONE BARRIER APART, so that while one
// row's four waves feed the matrix pipes the other row's four read LDS, convert int8 -> fp16 and issue the next tiles' DMA
// on the same SIMDs (two waves per SIMD in anti-phase instead of in lock step).
//   * 3-stage ring of 48 KiB (X 256 x 64 fp16 + W 256 x 64 int8), LDS-DMA with the chunk swizzles of k_gemm_dev.h on the source
//     address; tile t+2 is issued during phases 1..3 of tile t (6 pieces per wave), `s_waitcnt vmcnt(6)` in phase 3 retires tile
//     t+1 -- never a drain to zero inside the loop;
//   * reads follow the quadrant order: phase 1 X rows 0..63 of the wave (8 reads) + weight rows 32..63 (2 reads), phase 3 X rows
//     64..127 (8 reads), phase 4 the NEXT tile's weight rows 0..31 (2 reads: its buffer was retired in phase 3 and two barriers
//     lie in between); every fragment is read one phase before its conversion / MFMA, so `lgkmcnt(0)` after the barrier is free;
//   * a buffer is restaged two barriers or more after its last read (tile t-1's last reads are in its phase 3 -- phase 4 reads the
//     next buffer -- and tile t+2's first DMA into that buffer is issued in phase 1 of tile t).
// Numerics: identical to the other tile kernels (exact int8 -> fp16, fp32 accumulate, scale and one rounding in the epilogue).
#include <stdlib.h>

#include "k_gemm_dev.h"

namespace pplhip {

constexpr int E_BN = 256, E_BM = 256, E_ST = 3;
constexpr int E_XS = E_BM * G_BK * 2, E_WS = E_BN * G_BK, E_STAGE = E_XS + E_WS;  // 32 KiB + 16 KiB

#define E_BARRIER()                                 \
    do {                                            \
        asm volatile("" ::: "memory");              \
        __builtin_amdgcn_s_barrier();               \
        asm volatile("" ::: "memory");              \
    } while (0)

template <int EPI>
__global__ __launch_bounds__(512) void gemm_w8_8ph_kernel(const uint16_t* __restrict__ x, const int8_t* __restrict__ w,
                                                          const uint16_t* __restrict__ scale, int64_t M, int N, int K,
                                                          void* __restrict__ yv, int64_t ldy, int n_tiles, int m_tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem_e[];  // E_ST x (X | W)
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int nt = xcd + 8 * (slot / m_tiles);
    const int mt = slot % m_tiles;
    if (nt >= n_tiles) return;
    const int n0 = nt * E_BN;
    const int64_t m0 = (int64_t)mt * E_BM;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int wn = wave & 3, wm = wave >> 2;
    const int nb = wn * 64, mb = wm * 128;

    // ---- LDS-DMA sources (per lane; + k0 per tile) and wave-uniform destinations
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
    const uint32_t xdst = __builtin_amdgcn_readfirstlane(lds_addr(smem_e) + wave * 1024);
    const uint32_t wdst = xdst + E_XS;
    const int ktiles = K / G_BK;

    // ---- fragment offsets inside a stage (bytes)
    int woff[4], xoff[8][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = nb + i * 16 + l15;
        woff[i] = E_XS + row * G_BK + (kq ^ w_swz(row)) * 16;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = mb + j * 16 + l15;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) xoff[j][ks] = (row * G_BK + g_swz(row, ks * 4 + kq) * 8) * 2;
    }

    f4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    // prologue: tiles 0 and 1 in flight, tile 0 retired and published
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        if (d < ktiles) {
#pragma unroll
            for (int j = 0; j < 4; ++j) glds16(xsrc[j] + d * G_BK, xdst + d * E_STAGE + j * 8192);
#pragma unroll
            for (int j = 0; j < 2; ++j) glds16(wsrc[j] + d * G_BK, wdst + d * E_STAGE + j * 8192);
        }
    }
    if (ktiles > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    E_BARRIER();
    // raw int8 weight fragments (16 k values per lane); scalars, not arrays: a loop-carried array ends up in scratch memory
    uint4 w0 = *reinterpret_cast<const uint4*>(smem_e + woff[0]);
    uint4 w1 = *reinterpret_cast<const uint4*>(smem_e + woff[1]);
    if (wm == 1) E_BARRIER();  // the second wave row runs one barrier behind the first

    int cur = 0;  // stage of tile t
    for (int t = 0; t < ktiles; ++t) {
        const char* sb = smem_e + cur * E_STAGE;
        const int nxt = cur == E_ST - 1 ? 0 : cur + 1;
        const int st2 = nxt == E_ST - 1 ? 0 : nxt + 1;  // stage of tile t + 2 (= stage of tile t - 1)
        const bool pre = t + 2 < ktiles;
        const int k2 = (t + 2) * G_BK;
        h8 a01[2][2], a23[2][2], xb[4][2];

        // ---------------- phase 1: quadrant (weight rows 0..31) x (X rows 0..63)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) xb[j][ks] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(sb + xoff[j][ks]));
        const uint4 w2 = *reinterpret_cast<const uint4*>(sb + woff[2]);
        const uint4 w3 = *reinterpret_cast<const uint4*>(sb + woff[3]);
        a01[0][0] = cvt_i8x8_f16(make_uint2(w0.x, w0.y));
        a01[0][1] = cvt_i8x8_f16(make_uint2(w0.z, w0.w));
        a01[1][0] = cvt_i8x8_f16(make_uint2(w1.x, w1.y));
        a01[1][1] = cvt_i8x8_f16(make_uint2(w1.z, w1.w));
        if (pre) {
            glds16(xsrc[0] + k2, xdst + st2 * E_STAGE);
            glds16(xsrc[1] + k2, xdst + st2 * E_STAGE + 8192);
        }
        E_BARRIER();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a01[i][ks], xb[j][ks], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        E_BARRIER();

        // ---------------- phase 2: (weight rows 32..63) x (X rows 0..63)
        a23[0][0] = cvt_i8x8_f16(make_uint2(w2.x, w2.y));
        a23[0][1] = cvt_i8x8_f16(make_uint2(w2.z, w2.w));
        a23[1][0] = cvt_i8x8_f16(make_uint2(w3.x, w3.y));
        a23[1][1] = cvt_i8x8_f16(make_uint2(w3.z, w3.w));
        if (pre) {
            glds16(xsrc[2] + k2, xdst + st2 * E_STAGE + 2 * 8192);
            glds16(xsrc[3] + k2, xdst + st2 * E_STAGE + 3 * 8192);
        }
        E_BARRIER();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[2 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a23[i][ks], xb[j][ks], acc[2 + i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        E_BARRIER();

        // ---------------- phase 3: (weight rows 32..63) x (X rows 64..127); retires tile t + 1
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) xb[j][ks] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(sb + xoff[4 + j][ks]));
        if (pre) {
            glds16(wsrc[0] + k2, wdst + st2 * E_STAGE);
            glds16(wsrc[1] + k2, wdst + st2 * E_STAGE + 8192);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // all but tile t + 2's six pieces: tile t + 1 has landed
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        E_BARRIER();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[2 + i][4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a23[i][ks], xb[j][ks], acc[2 + i][4 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        E_BARRIER();

        // ---------------- phase 4: (weight rows 0..31) x (X rows 64..127); the next tile's first weight fragments
        uint4 wn0 = w0, wn1 = w1;
        if (t + 1 < ktiles) {
            wn0 = *reinterpret_cast<const uint4*>(smem_e + nxt * E_STAGE + woff[0]);
            wn1 = *reinterpret_cast<const uint4*>(smem_e + nxt * E_STAGE + woff[1]);
        }
        E_BARRIER();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a01[i][ks], xb[j][ks], acc[i][4 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        w0 = wn0;
        w1 = wn1;
        E_BARRIER();
        cur = nxt;
    }
    if (wm == 0) E_BARRIER();  // barrier counts of the two wave rows match again

#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + nb + i * 16 + kq * 4;
        if (n >= N) continue;
        const h4 sh = __builtin_bit_cast(h4, *reinterpret_cast<const uint2*>(scale + n));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t m = m0 + mb + j * 16 + l15;
            if (m >= M) continue;
            store4<EPI>(yv, ldy, m, n, acc[i][j][0] * (float)sh[0], acc[i][j][1] * (float)sh[1], acc[i][j][2] * (float)sh[2],
                        acc[i][j][3] * (float)sh[3]);
        }
    }
}

hipError_t launch_gemm_w8_8ph(hipStream_t s, const uint16_t* x, const int8_t* w, const uint16_t* scale, int64_t M, int N, int K, void* y,
                              int64_t ldy, int epi) {
    const int nt2 = (N + E_BN - 1) / E_BN, mt2 = (int)((M + E_BM - 1) / E_BM);
    const size_t lds = (size_t)E_ST * E_STAGE;
    dim3 grid((unsigned)((nt2 + 7) / 8 * 8 * mt2));
    static bool attr_dev[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_dev[dev & 63]) {
        (void)hipFuncSetAttribute((const void*)gemm_w8_8ph_kernel<EPI_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)gemm_w8_8ph_kernel<EPI_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)gemm_w8_8ph_kernel<EPI_SWIGLU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_dev[dev & 63] = true;
    }
#define L8(E) hipLaunchKernelGGL((gemm_w8_8ph_kernel<E>), grid, dim3(512), lds, s, x, w, scale, M, N, K, y, ldy, nt2, mt2)
    if (epi == EPI_F32) L8(EPI_F32); else if (epi == EPI_F16) L8(EPI_F16); else L8(EPI_SWIGLU);
#undef L8
    return hipGetLastError();
}

}  // namespace pplhip
