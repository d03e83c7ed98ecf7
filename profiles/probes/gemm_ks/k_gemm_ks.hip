// W8A16 tile GEMM for the per-rank SLICES of a tensor-parallel decode step (round 6): block tiles of 128 x 96 and 64 x 96 whose four
// multiplying waves split the 64-deep K tile BY K-STEP instead of by rows or columns.
//
// Why.  At tensor-parallel 8 the 7B layer's wqkv / w13 are N = 1536 / 2752 at M = 1024: 96 / 176 tiles of 128 x 128 on 256 CUs (a K split
// with fp32 slabs, or 31 % of the CUs idle) -- profiles/r06_tp8_slice_kernel_stats.csv: 22.2 + 33.6 us per layer = 0.58 / 0.69 PFLOP/s.
// Narrower tiles fix the count (64 x 96: 256 tiles; 128 x 96: 232) but not with the 16 x 16 x 32 wave tiles of gemm_dma_body: every
// multiplying wave re-reads the whole activation tile from LDS, and at 32 x 64 per wave the fragment reads (72 KiB per K tile and CU)
// outweigh the MFMAs.  Here each of the four multiplying waves (one per SIMD) owns the WHOLE block tile for ONE 16-deep k-step of every
// K tile on v_mfma_f32_32x32x16_f16: per K tile a wave reads MI activation + NI weight fragments (128 x 96: 5.5 KiB) for MI x NI MFMAs of
// 32 cycles -- 22 KiB of fragment reads per K tile and CU -- and converts each int8 weight once.  The four partial sums meet once, behind
// the K loop, through the (then idle) ring.  Fragments of tile t + 1 are read behind the barrier that publishes it while the MFMAs of
// tile t run from registers (one rolling register set, as in k_gemm_pc.hip), so a wave's LDS latency hides behind its own MFMAs.
// Four producer waves only issue the ring's LDS-DMA (k_gemm_wide.hip's scheme; surplus pieces re-load the last one so that every
// producer's vmcnt is static).
// LDS image and fragment addressing as in k_gemm_asm.hip: activations [rows][64 fp16], 16-byte chunk q of row r at position
// q ^ ((r >> 1) & 7); weights [rows][64 int8], chunk c of row r at position c ^ ((r >> 2) & 3).  MFMA k-step s: lane (r = l & 31,
// h = l >> 5) multiplies k = 16 s + 8 h .. + 8 of both operands; accumulator value e of lane (r, h) = channel 8 (e >> 2) + 4 h + (e & 3)
// of the 32-channel block, activation row r of the 32-row block.
// Oracle: ref_linear_raw (oracle/llama_ref.c); per output the fp32 sum runs over k = s (mod 4 k-steps) per wave, then over the 4 waves.
#include <stdlib.h>
#include "k_gemm_dev.h"

namespace pplhip {

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));
#ifndef KS_ABL   // diagnosis builds only (make EXTRA=-DKS_ABL=n; WRONG results): 1 no MFMAs, 2 no ring refills, 4 no fragment re-reads, 8 no loop barriers
#define KS_ABL 0
#endif
constexpr int KS_NP = 4, KS_NC = 4;   // producer waves, multiplying waves (= k-steps of a 64-deep K tile)

template <int EPI, int MI, int NI, int ST>
__global__ __launch_bounds__((KS_NC + KS_NP) * 64) void gemm_w8_ks_kernel(const uint16_t* __restrict__ x, const int8_t* __restrict__ w,
                                                                          const uint16_t* __restrict__ scale, int64_t M, int N, int K,
                                                                          void* __restrict__ yv, int64_t ldy, int n_tiles, int m_tiles) {
    constexpr int BM = 32 * MI, BN = 32 * NI;
    constexpr int XB = BM * G_BK * 2, WB = BN * G_BK;          // bytes per stage
    constexpr int XP = XB / 1024, WP = WB / 1024;              // one-KiB DMA pieces per stage
    constexpr int PP = (XP + WP + KS_NP - 1) / KS_NP;          // pieces per producer wave and tile (surplus: duplicates of the last piece)
    constexpr int NA = MI * NI;                                // accumulators per wave
    extern __shared__ __attribute__((aligned(128))) char smem_ks[];   // ST x activations, then ST x weights; reused for the final exchange
    char* const Xs0 = smem_ks;
    char* const Wq0 = smem_ks + ST * XB;

    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;   // XCD id % 8 owns the weight tiles n == xcd (mod 8) and all their m tiles
    const int nt = xcd + 8 * (slot / m_tiles);
    const int mt = slot % m_tiles;
    if (nt >= n_tiles) return;
    const int n0 = nt * BN;
    const int64_t m0 = (int64_t)mt * BM;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ktiles = K / G_BK;

    // Ring protocol (both roles): the barrier of iteration t publishes tile t + 1 and frees the stage of tile t (its fragments went to
    // registers during iteration t - 1); behind it the producers refill that stage with tile t + ST.  Tiles 0 .. ST - 1 are issued up front.
    if (wave >= KS_NC) {
        const int pw = wave - KS_NC;
        const char* psrc[PP];
        uint32_t pdst[PP];
        int pstep[PP], pstage[PP];
        const uint32_t xbase = lds_addr(Xs0), wbase = lds_addr(Wq0);
#pragma unroll
        for (int j = 0; j < PP; ++j) {
            int P = pw + KS_NP * j;
            if (P >= XP + WP) P = XP + WP - 1;                 // surplus piece: the last weight piece again (same bytes, same place)
            if (P < XP) {
                const int p = P * 64 + lane, row = p >> 3, c = (p & 7) ^ ((row >> 1) & 7);
                int64_t m = m0 + row;
                if (m >= M) m = M - 1;
                psrc[j] = reinterpret_cast<const char*>(x + m * K + c * 8);
                pdst[j] = __builtin_amdgcn_readfirstlane(xbase + P * 1024);
                pstep[j] = G_BK * 2; pstage[j] = XB;
            } else {
                const int Pw = P - XP;
                const int p = Pw * 64 + lane, row = p >> 2, c = (p & 3) ^ ((row >> 2) & 3);
                int n = n0 + row;
                if (n >= N) n = N - 1;
                psrc[j] = reinterpret_cast<const char*>(w) + (int64_t)n * K + c * 16;
                pdst[j] = __builtin_amdgcn_readfirstlane(wbase + Pw * 1024);
                pstep[j] = G_BK; pstage[j] = WB;
            }
        }
#define KS_PRODUCE(KT, STG)                                                                                     \
    do {                                                                                                        \
        _Pragma("unroll") for (int j = 0; j < PP; ++j)                                                          \
            glds16(psrc[j] + (int64_t)(KT) * pstep[j], pdst[j] + (STG) * pstage[j]);                            \
    } while (0)
#pragma unroll
        for (int d = 0; d < ST; ++d)
            if (d < ktiles) KS_PRODUCE(d, d);
        // tile 0 landed -> the consumers' first fragment reads
        {
            const int younger = ktiles - 1 < ST - 1 ? ktiles - 1 : ST - 1;
            if (younger >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PP) : "memory");
            else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PP) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PP) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        int stn = 0;   // stage of tile t (= the one refilled behind barrier t)
        for (int t = 0; t < ktiles; ++t) {
            // tile t + 1 must have landed: younger tiles in flight = min(ktiles - 2 - t, ST - 2)
            int younger = ktiles - 2 - t;
            if (younger > ST - 2) younger = ST - 2;
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PP) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PP) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!(KS_ABL & 8)) __syncthreads();
            if (t + ST < ktiles && !(KS_ABL & 2)) KS_PRODUCE(t + ST, stn);
            stn = stn == ST - 1 ? 0 : stn + 1;
        }
#undef KS_PRODUCE
        // the consumers' exchange barriers
#pragma unroll
        for (int g = 0; g < (NA + 3) / 4; ++g) { __syncthreads(); __syncthreads(); }
        return;
    }

    // ---- multiplying wave `wave` = k-step `wave` of every K tile ------------------------------------------------------------------
    static_assert(ST == 4, "the producers' wait counts above assume a 4-stage ring");
    const int r = lane & 31, h = lane >> 5;
    const uint32_t xoff = r * 128 + (((2 * wave + h) ^ ((r >> 1) & 7)) << 4);
    const uint32_t woff = r * 64 + ((wave ^ ((r >> 2) & 3)) << 4) + h * 8;
    f16v acc[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[a][k] = 0.f;
    h8 xa[MI], wa[NI];
    uint2 wraw[NI];
    __syncthreads();   // tile 0 is published
#pragma unroll
    for (int j = 0; j < MI; ++j) xa[j] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(Xs0 + j * 4096 + xoff));
#pragma unroll
    for (int i = 0; i < NI; ++i) wa[i] = cvt_i8x8_f16(*reinterpret_cast<const uint2*>(Wq0 + i * 2048 + woff));
    int st = 1 % ST;   // stage of tile t + 1
    for (int t = 0; t < ktiles; ++t) {
        if (!(KS_ABL & 8)) __syncthreads();   // tile t + 1 is published (behind the last tile the reads below fetch a stale stage and the result is dropped: a
                           // branch here would make hipcc wait for the reads in front of the MFMAs, at the block join)
        const char* xs = Xs0 + st * XB + xoff;
        const char* ws = Wq0 + st * WB + woff;
        st = st == ST - 1 ? 0 : st + 1;
        // ONE rolling register set: the raw weights of tile t + 1 are requested first; column j of the activation fragments is re-read for
        // tile t + 1 right behind the NI MFMAs that were its last readers in tile t; the weights are converted behind the last MFMA.  The
        // order is pinned (sched_group_barrier): left alone, hipcc sinks every read to just in front of its first use in the next iteration
        // and the wave waits out the LDS latency there, with no other multiplying wave on its SIMD to fill the gap.
#pragma unroll
        for (int i = 0; i < NI; ++i) if (!(KS_ABL & 4) || t == 0) wraw[i] = *reinterpret_cast<const uint2*>(ws + i * 2048);
        __builtin_amdgcn_sched_group_barrier(0x100, NI, 0);          // DS reads: the raw weights
#pragma unroll
        for (int j = 0; j < MI; ++j) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                if (KS_ABL & 1) acc[i * MI + j][0] += (float)wa[i][0] * (float)xa[j][0];
                else acc[i * MI + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[i], xa[j], acc[i * MI + j], 0, 0, 0);
            }
            if (!(KS_ABL & 4) || t == 0) xa[j] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(xs + j * 4096));
            __builtin_amdgcn_sched_group_barrier(0x008, NI, 0);      // NI MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // one DS read
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) wa[i] = cvt_i8x8_f16(wraw[i]);
    }

    // ---- the four k-step partial sums meet: accumulator a belongs to wave a & 3; groups of four accumulators go through LDS
    // [slot = 3 x owner + (source wave's rank among the owner's three peers)][e / 4][lane] float4 = 12 KiB per owner
    float4* const red = reinterpret_cast<float4*>(smem_ks);
#pragma unroll
    for (int g = 0; g < (NA + 3) / 4; ++g) {
        __syncthreads();   // the buffer is free (first group: every wave is past its last fragment read)
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int a = 4 * g + o;
            if (a < NA && o != wave) {
                const int slot = 3 * o + (wave < o ? wave : wave - 1);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    red[(slot * 4 + q) * 64 + lane] = make_float4(acc[a][4 * q], acc[a][4 * q + 1], acc[a][4 * q + 2], acc[a][4 * q + 3]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int a = 4 * g + o;
            if (a < NA && o == wave) {
#pragma unroll
                for (int p = 0; p < 3; ++p)     // peers in wave order: one fixed summation order
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = red[((3 * o + p) * 4 + q) * 64 + lane];
                        acc[a][4 * q] += v.x; acc[a][4 * q + 1] += v.y; acc[a][4 * q + 2] += v.z; acc[a][4 * q + 3] += v.w;
                    }
            }
        }
    }
    // ---- epilogue: wave `wave` stores the accumulators it owns
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        if ((a & 3) != wave) continue;
        const int i = a / MI, j = a % MI;
        const int64_t m = m0 + 32 * j + r;
        if (m >= M) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = n0 + 32 * i + 8 * q + 4 * h;
            if (n >= N) continue;
            const h4 sh = __builtin_bit_cast(h4, *reinterpret_cast<const uint2*>(scale + n));
            store4<EPI>(yv, ldy, m, n, acc[a][4 * q] * (float)sh[0], acc[a][4 * q + 1] * (float)sh[1], acc[a][4 * q + 2] * (float)sh[2],
                        acc[a][4 * q + 3] * (float)sh[3]);
        }
    }
}

constexpr int KS_ST = 4;
template <int MI, int NI>
constexpr int ks_lds_bytes() {
    constexpr int ring = KS_ST * (32 * MI * G_BK * 2 + 32 * NI * G_BK), red = 4 * 3 * 4 * 64 * 16;   // ring; exchange buffer (48 KiB)
    return ring > red ? ring : red;
}

template <int EPI, int MI, int NI>
hipError_t ks_launch(hipStream_t s, const uint16_t* x, const int8_t* w, const uint16_t* scale, int64_t M, int N, int K, void* y, int64_t ldy) {
    constexpr int lds = ks_lds_bytes<MI, NI>();
    const int n_tiles = (N + 32 * NI - 1) / (32 * NI), m_tiles = (int)((M + 32 * MI - 1) / (32 * MI));
    static bool attr_dev[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_dev[dev & 63]) {
        (void)hipFuncSetAttribute((const void*)gemm_w8_ks_kernel<EPI, MI, NI, KS_ST>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_dev[dev & 63] = true;
    }
    dim3 grid((unsigned)((n_tiles + 7) / 8 * 8 * m_tiles));
    hipLaunchKernelGGL((gemm_w8_ks_kernel<EPI, MI, NI, KS_ST>), grid, dim3((KS_NC + KS_NP) * 64), lds, s, x, w, scale, M, N, K, y, ldy, n_tiles, m_tiles);
    return hipGetLastError();
}

}  // namespace

// Which k-split tile (0: none, 1: 128 x 96, 2: 64 x 96) beats the 128 x 128 grid for this shape: a slice whose 128 x 128 tiles leave a
// quarter of the CUs idle (or need K slabs) while the narrower tiles fill ONE round of 256 blocks.  512 <= M <= 1024 (the decode steps
// the slices were measured at), K long enough that the final exchange (~1 us) is noise.
int linear_w8_ks_tile(int64_t M, int N, int K) {
    static const int mode = getenv("PPLHIP_GEMM_KS") ? atoi(getenv("PPLHIP_GEMM_KS")) : 1;   // 0: off (A/B runs)
    if (!mode || M < 512 || M > 1024 || K % G_BK || K < 1024 || N % 4) return 0;
    const int64_t t128 = (int64_t)((N + 127) / 128) * ((M + 127) / 128);
    if (t128 > 256) return 0;
    const double eff128 = (double)M * N / (256.0 * 128 * 128);
    int best = 0;
    double beff = eff128 + 0.12;    // the new tile must fill the chip clearly better
    const int64_t ta = (int64_t)((N + 95) / 96) * ((M + 127) / 128), tb = (int64_t)((N + 95) / 96) * ((M + 63) / 64);
    if (ta <= 256) { const double e = (double)M * N / (256.0 * 128 * 96); if (e > beff) { beff = e; best = 1; } }
    if (tb <= 256) { const double e = (double)M * N / (256.0 * 64 * 96); if (e > beff) { beff = e; best = 2; } }
    return best;
}

hipError_t launch_linear_w8_ks(hipStream_t s, const uint16_t* x, const int8_t* w, const uint16_t* scale, int64_t M, int N, int K, void* y,
                               int64_t ldy, int epi, int tile) {
    if (K % G_BK || N % 4 || ldy % 4 || (tile != 1 && tile != 2) || (epi != EPI_F16 && epi != EPI_SWIGLU)) return hipErrorInvalidValue;
    if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)scale & 7)) return hipErrorInvalidValue;
    if (tile == 1) return epi == EPI_F16 ? ks_launch<EPI_F16, 4, 3>(s, x, w, scale, M, N, K, y, ldy) : ks_launch<EPI_SWIGLU, 4, 3>(s, x, w, scale, M, N, K, y, ldy);
    return epi == EPI_F16 ? ks_launch<EPI_F16, 2, 3>(s, x, w, scale, M, N, K, y, ldy) : ks_launch<EPI_SWIGLU, 2, 3>(s, x, w, scale, M, N, K, y, ldy);
}

}  // namespace pplhip
