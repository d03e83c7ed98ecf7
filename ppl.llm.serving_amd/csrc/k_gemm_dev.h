// Device side of the tile GEMM (128 x 128 x 64, LDS-DMA ring): constants, epilogues, conversion helpers and the block body
// gemm_dma_body; k_gemm.hip wraps it into the kernels and k_gemm_i8.hip shares the helpers.
#pragma once
#include <type_traits>
#include "kernels.h"

namespace pplhip {

constexpr int G_BN = 128, G_BM = 128, G_BK = 64;

__device__ __forceinline__ int g_swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// Epilogues: EPI 0 = fp16 output, 1 = fp32 output (logits), 2 = fused SwiGLU -- the weight rows are stored interleaved
// (gate_0, up_0, gate_1, up_1, ...), so the 4 consecutive output channels a lane owns are two (gate, up) pairs and the
// lane writes silu(gate) * up for both: y is then [M, N/2] (K10 fused into the producing GEMM; numerics as the separate
// kernel: gate and up are rounded to fp16 first).
enum { EPI_F16 = 0, EPI_F32 = 1, EPI_SWIGLU = 2 };
template <int EPI>
__device__ __forceinline__ void store4(void* yv, int64_t ldy, int64_t m, int n, float v0, float v1, float v2, float v3) {
    if constexpr (EPI == EPI_F32) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(yv) + m * ldy + n) = make_float4(v0, v1, v2, v3);
    } else if constexpr (EPI == EPI_F16) {
        const h4 o = {to_h(v0), to_h(v1), to_h(v2), to_h(v3)};
        *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(yv) + m * ldy + n) = __builtin_bit_cast(uint2, o);
    } else {
        const float g0 = round_h(v0), u0 = round_h(v1), g1 = round_h(v2), u1 = round_h(v3);
        const h2 o = {to_h(g0 / (1.0f + __expf(-g0)) * u0), to_h(g1 / (1.0f + __expf(-g1)) * u1)};
        *reinterpret_cast<uint32_t*>(reinterpret_cast<uint16_t*>(yv) + m * ldy + (n >> 1)) = __builtin_bit_cast(uint32_t, o);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// W8A16 fast path (K % 64 == 0): both operands go global -> LDS by DMA (global_load_lds_dwordx4: no staging VGPRs, no
// ds_write pass), the weight tile stays int8 in LDS (half the LDS bytes and half the fragment-read bytes) and is
// converted to fp16 in registers right before the MFMA (exact: x ^ 0x80 -> 0x6400 | u8 -> minus 1152, two VALU ops per
// pair, in the shadow of the matrix pipe).  Two LDS stages, one barrier per K tile: the DMA of tile t+1 is issued
// right after the barrier that publishes tile t and lands during the MFMAs of tiles t and t+1 (3-stage ring).
// The DMA writes LDS linearly (wave-uniform base + lane * 16 B), so the XOR swizzle is applied to the per-lane SOURCE
// address and again on the fragment reads (cdna_hip_programming.md rule 21).
// ---------------------------------------------------------------------------------------------------------------
// 16-byte chunk swizzle of the int8 weight tile (64-byte rows, 4 rows per 256-byte bank window): conflict-free for
// the four 16-lane groups ds_read_b128 is serviced in ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ... -- MI355X_MICROARCH LDS)
__device__ __forceinline__ int w_swz(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }

// 8 nibbles (low nibble = even k, value = nibble - 8) -> 8 fp16 in k order, times the group scale: split the even and
// odd nibbles into bytes, v_perm pairs them up under the fp16 exponent 0x64 (1024 + n, exact), subtract 1032, scale.
__device__ __forceinline__ h8 cvt_i4x8_f16(uint32_t w, h2 sc) {
    const uint32_t ev = w & 0x0f0f0f0fu, od = (w >> 4) & 0x0f0f0f0fu;  // bytes: k = 0,2,4,6 / 1,3,5,7
    const h2 bias = {(_Float16)1032.0f, (_Float16)1032.0f};
    const h2 a = (__builtin_bit_cast(h2, __builtin_amdgcn_perm(od, ev, 0x0c040c00u) | 0x64006400u) - bias) * sc;
    const h2 b = (__builtin_bit_cast(h2, __builtin_amdgcn_perm(od, ev, 0x0c050c01u) | 0x64006400u) - bias) * sc;
    const h2 c = (__builtin_bit_cast(h2, __builtin_amdgcn_perm(od, ev, 0x0c060c02u) | 0x64006400u) - bias) * sc;
    const h2 d = (__builtin_bit_cast(h2, __builtin_amdgcn_perm(od, ev, 0x0c070c03u) | 0x64006400u) - bias) * sc;
    return h8{a[0], a[1], b[0], b[1], c[0], c[1], d[0], d[1]};
}

// LDS-DMA issued from inline asm so that hipcc does not see an outstanding LDS write and does not drain vmcnt(0)
// in front of the fragment reads of the CURRENT tile (cdna_hip_programming.md 5.7: M0 is written and restored in the
// same statement; completion is waited for by hand with s_waitcnt vmcnt(0) before the publishing barrier).
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_wave_base /* wave-uniform LDS byte address */) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_wave_base)
                 : "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}

constexpr int W4_MAXG = 64;  // W4: quantisation groups (of 128 along K) whose scales one block keeps in LDS

// WL = wave layout of the 128x128x64 block tile:
//   0: 4 waves, 2x2, each 64(n) x 64(m);  1: 4 waves, 4x1, each 32(n) x 128(m) x k64;
//   5: 8 waves: 4 consumers with the layout of 1 and 4 producers that do nothing but issue the ring's LDS-DMA;
//   6: 12 waves: 8 consumers -- the four of 5 twice, group h multiplying only k-step h of every K tile (two consumer waves per SIMD whose
//      read / convert / MFMA chains overlap), partial sums added through LDS at the end -- and 4 producers; needs G_ST >= 3;
// WQ = 8: int8 weights + per-channel scale; WQ = 0: fp16; WQ = 4: int4, group 128
// LDS bytes gemm_dma_body needs
// MAXG (W4): quantisation groups whose scales the block keeps in LDS -- W4_MAXG, or 16 for launches whose K slabs span at most 32 tiles
// (44 instead of 56 KiB with two stages: three blocks per CU instead of two)
template <int WQ, int G_ST, int BM = G_BM, int MAXG = W4_MAXG>
constexpr int gemm_dma_lds_bytes() {
    constexpr int WB2 = WQ == 8 ? 2 : (WQ == 4 ? 1 : 4);
    return G_ST * (BM * G_BK * 2 + G_BN * G_BK * WB2 / 2) + (WQ == 4 ? MAXG * G_BN * 2 : 0);
}

// BM = 64 (WL 1 only): half-height tile for 16 < M <= 64 -- half the activation bytes per stage and half the MFMAs of a 128-row tile whose
// upper half would only repeat row M - 1
template <int WQ, int EPI, int G_ST, int WL, int BM = G_BM>
__device__ __forceinline__ void gemm_dma_body(const uint16_t* __restrict__ x, const void* __restrict__ wv,
                                              const uint16_t* __restrict__ scale, int64_t M, int N, int K, void* __restrict__ yv,
                                              int64_t ldy, int n_tiles, int m_tiles, int map_mode, int kt_per_split,
                                              float* __restrict__ ws, int block_id, int split_id, int n_splits, char* smem) {
    // (the caller declares ONE __shared__ object: a second one makes hipcc wait vmcnt(0) before every ds_read of a DMA pipeline)
    constexpr int WB2 = WQ == 8 ? 2 : (WQ == 4 ? 1 : 4);   // half-bytes per weight element
    constexpr int W_STAGE = G_BN * G_BK * WB2 / 2;         // 8 KiB (int8) / 16 KiB (fp16) / 4 KiB (int4)
    constexpr int NT = 256;                                // threads that move tile pieces (WL 5: the 4 producer waves)
    static_assert(BM == G_BM || (BM == 64 && WL == 1), "half-height tiles: 4-wave layout only");
    constexpr int X_DMA = BM * G_BK * 2 / (NT * 16);       // DMA instructions per wave per tile for X (4; 2 at BM = 64)
    constexpr int W_DMA = W_STAGE / (NT * 16);             // ... for W (2 int8, 4 fp16, 1 int4)
    // (W4: W4_MAXG * G_BN * 2 more bytes for the group scales of this block's rows, [group][row])
    // smem: G_ST * (G_BM * G_BK * 2 + W_STAGE) + SC_BYTES bytes (per stage: X 16 KiB + W), 16-byte aligned, provided by the caller
    uint16_t* const Xs0 = reinterpret_cast<uint16_t*>(smem);
    char* const Wq0 = smem + G_ST * BM * G_BK * 2;
    uint16_t* const Sc = reinterpret_cast<uint16_t*>(smem + G_ST * (BM * G_BK * 2 + W_STAGE));
    const char* w = reinterpret_cast<const char*>(wv);

    // block -> tile.  map_mode 1 (m_tiles % 8 == 0): XCD x = id % 8 owns the activation row-tiles m == x (mod 8) and
    // walks all weight tiles, so its 4 MiB L2 keeps that 1 MiB activation slice resident for the whole GEMM and the
    // weights stream through once per XCD.  map_mode 0: XCD x owns weight tiles n == x (mod 8) and walks all m tiles.
    const int id = block_id;
    const int xcd = id & 7, slot = id >> 3;
    int nt, mt;
    if ((map_mode & 0xff) == 1) {
        const int mg = m_tiles >> 3;
        mt = xcd + 8 * (slot % mg);
        nt = slot / mg;
    } else {
        // super-tiles of gn (n) x gm (m) tiles, m fastest inside, super-tiles m-major (gm = m_tiles: the plain n-major walk).  The ~96 blocks
        // an XCD runs at a time then share gn weight and gm activation tiles per K step (k_gemm.hip, gemm_w8_dma256_kernel)
        const int gm = (map_mode >> 16) & 0xff, gn = (map_mode >> 24) & 0xff;
        if (gm == 0) {
            nt = xcd + 8 * (slot / m_tiles);
            mt = slot % m_tiles;
        } else {
            const int per_super = gn * gm, sm_count = m_tiles / gm;
            const int sup = slot / per_super, within = slot % per_super;
            nt = xcd + 8 * ((sup / sm_count) * gn + within / gm);
            mt = (sup % sm_count) * gm + within % gm;
        }
    }
    if (nt >= n_tiles) return;
    const int n0 = nt * G_BN;
    const int64_t m0 = (int64_t)mt * BM;

    // WL 5: waves 4..7 only issue the LDS-DMA of the ring (producers), waves 0..3 only multiply (consumers, layout of WL 1)
    constexpr bool PC = WL == 5 || WL == 6;  // producer / consumer forms
    const bool producer = PC && threadIdx.x >= (WL == 6 ? 512 : 256);
    const int tid = PC ? (threadIdx.x & 255) : threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kh = WL == 6 ? (int)((threadIdx.x >> 8) & 1) : 0;  // WL 6: the k-step of every tile this consumer multiplies
    const int l15 = lane & 15, kq = lane >> 4;
    // WL 1: every wave owns 32 weight rows and all 128 activation rows of the tile, so each weight fragment is converted
    // (int8/int4 -> fp16) by exactly one wave and feeds 8 MFMAs; WL 0 converts every fragment in two waves for 4 MFMAs
    constexpr int NI = WL ? 2 : 4, NJ = WL ? BM / 16 : 4;
    const int wn = WL ? wave : wave >> 1, wm = WL ? 0 : wave & 1;
    const int nb = wn * (NI * 16), mb = wm * (NJ * 16);  // row bases of this wave inside the tile

    // per-lane DMA sources (constant over K except for the k0 term)
    const uint16_t* xsrc[X_DMA];
    const char* wsrc[W_DMA];
#pragma unroll
    for (int j = 0; j < X_DMA; ++j) {
        // LDS position pos = ks*4 + kq of a row holds the source chunk kq*2 + ks, i.e. k = kq*16 + ks*8 .. +8: lane
        // (ks, kq) then multiplies exactly the k range that ONE 16-byte read of the int8 weight row (chunk kq) delivers
        const int p = j * NT + tid, row = p >> 3, pos = (p & 7) ^ ((row >> 1) & 7);
        const int c = ((pos & 3) << 1) | (pos >> 2);
        int64_t m = m0 + row;
        if (m >= M) m = M - 1;  // rows past M are never stored
        xsrc[j] = x + m * K + c * 8;
    }
#pragma unroll
    for (int j = 0; j < W_DMA; ++j) {
        const int p = j * NT + tid;
        if constexpr (WQ == 8) {
            const int row = p >> 2, c = (p & 3) ^ w_swz(row);
            int n = n0 + row;
            if (n >= N) n = N - 1;
            wsrc[j] = w + (int64_t)n * K + c * 16;
        } else if constexpr (WQ == 4) {
            // 32-byte rows (64 nibbles): lane (row, kq) later reads bytes [kq*8, kq*8+8) = its k range kq*16 .. +16;
            // rows with bit 3 set swap their 16-byte halves so that rows r and r+8 do not share LDS banks
            const int row = p >> 1, c = (p & 1) ^ ((row >> 3) & 1);
            int n = n0 + row;
            if (n >= N) n = N - 1;
            wsrc[j] = w + ((int64_t)n * K) / 2 + c * 16;
        } else {  // fp16 weights: staged exactly like the activation tile
            const int row = p >> 3, pos = (p & 7) ^ ((row >> 1) & 7);
            const int c = ((pos & 3) << 1) | (pos >> 2);
            int n = n0 + row;
            if (n >= N) n = N - 1;
            wsrc[j] = w + ((int64_t)n * K + c * 8) * 2;
        }
    }
    // wave-uniform LDS destinations (byte addresses): piece j of this wave starts at (j * 256 + wave * 64) * 16
    const uint32_t xdst = __builtin_amdgcn_readfirstlane(lds_addr(Xs0) + wave * 1024);
    const uint32_t wdst = __builtin_amdgcn_readfirstlane(lds_addr(Wq0) + wave * 1024);
    auto issue = [&](int stage, int k0) {
#pragma unroll
        for (int j = 0; j < X_DMA; ++j) glds16(xsrc[j] + k0, xdst + stage * (BM * G_BK * 2) + j * (NT * 16));
#pragma unroll
        for (int j = 0; j < W_DMA; ++j) glds16(wsrc[j] + k0 * WB2 / 2, wdst + stage * W_STAGE + j * (NT * 16));
    };

    f4 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    // G_ST-stage ring, prefetch distance D = G_ST - 1: while tile t is multiplied, tiles t+1 .. t+D are in flight.  Every
    // wave issues 6 DMA instructions per tile, so "all but the newest 6*j have landed" (vmcnt(6*j)) == tile t is complete
    // when j younger tiles have been issued.
    constexpr int D = G_ST - 1;
    // split-K (small M): split `split_id` owns K tiles [kt0, kt0 + ktiles) and writes an fp32 partial slab; a reduce kernel
    // sums the slabs, applies the channel scales and rounds
    const int kt_all = K / G_BK;
    const int kt0 = split_id * kt_per_split;
    const int ktiles = (kt0 + kt_per_split < kt_all) ? kt_per_split : kt_all - kt0;
    if constexpr (WQ == 4) {
        // group scales of this block's 128 rows over its K range -> LDS, transposed to [group][row] (one 32-byte run per
        // 16-lane fragment read); kt0 and kt_per_split are even (launcher), so a group never straddles two splits
        const int G = K / 128, g0 = kt0 >> 1, ng = (ktiles + 1) >> 1;
        for (int e = tid; e < ng * G_BN; e += NT) {
            const int row = e / ng, g = e - row * ng;
            int n = n0 + row;
            if (n >= N) n = N - 1;
            Sc[g * G_BN + row] = scale[(int64_t)n * G + g0 + g];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // before the counted DMA pipeline starts
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
        if (d < ktiles && (!PC || producer)) issue(d, (kt0 + d) * G_BK);
    }
    int st = 0, stn = D;  // stage of tile t, stage of tile t+D
    h2 gsc[NI];           // W4: this lane's row scales of the current group
    if (PC && producer) {
        // producer waves: wait for their own pieces of tile t, meet the consumers at the barrier, refill the freed stage.
        // The LDS-DMA issue (~100 cycles per piece for the issuing wave) now runs beside the consumers' MFMA stream on
        // the SIMD instead of in front of it.
        for (int t = 0; t < ktiles; ++t) {
            const int younger = (ktiles - 1 - t) < (D - 1) ? (ktiles - 1 - t) : (D - 1);
            constexpr int PT = X_DMA + W_DMA;
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PT) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t + D < ktiles) issue(stn, (kt0 + t + D) * G_BK);
            stn = stn == G_ST - 1 ? 0 : stn + 1;
        }
        if constexpr (WL == 6) { __syncthreads(); __syncthreads(); }  // the consumers' two barriers around the partial-sum exchange
        return;
    }
    for (int t = 0; t < ktiles; ++t) {
        const int younger = (ktiles - 1 - t) < (D - 1) ? (ktiles - 1 - t) : (D - 1);
        if constexpr (!PC) {   // every wave issues PT = X_DMA + W_DMA DMA instructions per tile
            constexpr int PT = X_DMA + W_DMA;
            if (map_mode & 0x200) {}  // ablation 2 (PPLHIP_GEMM_ABLATE=2, wrong results): the DMA is issued but never waited for
            else if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PT) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();  // tile t is published; the stage read during iteration t-1 (== stage of tile t+D) is free
        if constexpr (!PC) if (t + D < ktiles && !(map_mode & 0x100)) issue(stn, (kt0 + t + D) * G_BK);  // 0x100: ablation (PPLHIP_GEMM_ABLATE)
        const uint16_t* xs = Xs0 + st * (BM * G_BK);
        const char* wq = Wq0 + st * W_STAGE;
        st = st == G_ST - 1 ? 0 : st + 1;
        stn = stn == G_ST - 1 ? 0 : stn + 1;
        // one 16-byte read per weight row delivers the int8 operands of BOTH k-steps of this lane (k = kq*16 .. +16)
        uint4 wraw[NI];
        if constexpr (WQ == 8) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int row = nb + i * 16 + l15;
                wraw[i] = *reinterpret_cast<const uint4*>(&wq[row * G_BK + (kq ^ w_swz(row)) * 16]);
            }
        } else if constexpr (WQ == 4) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int row = nb + i * 16 + l15;
                const uint2 v = *reinterpret_cast<const uint2*>(&wq[row * 32 + ((kq ^ (((row >> 3) & 1) << 1)) * 8)]);
                wraw[i].x = v.x; wraw[i].y = v.y;
            }
            if ((t & 1) == 0) {
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const _Float16 sv = __builtin_bit_cast(_Float16, Sc[(t >> 1) * G_BN + nb + i * 16 + l15]);
                    gsc[i] = h2{sv, sv};
                }
            }
        }
#pragma unroll
        for (int kx = 0; kx < (WL == 6 ? 1 : 2); ++kx) {
            const int ks = WL == 6 ? kh : kx;
            h8 a[NI], bfr[NJ];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                if constexpr (WQ == 8) {
                    a[i] = cvt_i8x8_f16(ks == 0 ? make_uint2(wraw[i].x, wraw[i].y) : make_uint2(wraw[i].z, wraw[i].w));
                } else if constexpr (WQ == 4) {
                    a[i] = cvt_i4x8_f16(ks == 0 ? wraw[i].x : wraw[i].y, gsc[i]);
                } else {
                    const int row = nb + i * 16 + l15;
                    a[i] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(&wq[(row * G_BK + g_swz(row, ks * 4 + kq) * 8) * 2]));
                }
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int row = mb + j * 16 + l15;
                bfr[j] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(&xs[row * G_BK + g_swz(row, ks * 4 + kq) * 8]));
            }
#ifdef GEMM_SETPRIO
            __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], bfr[j], acc[i][j], 0, 0, 0);
#ifdef GEMM_SETPRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        }
    }

    if constexpr (WL == 6) {
        // the k-step-1 group hands its partial sums to the k-step-0 group through the (now idle) ring: [wave][i][j][lane] float4
        float4* red = reinterpret_cast<float4*>(smem);
        __syncthreads();  // every consumer is past its last fragment read
        if (kh == 1) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    red[((wave * NI + i) * NJ + j) * 64 + lane] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
        __syncthreads();
        if (kh == 1) return;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float4 o = red[((wave * NI + i) * NJ + j) * 64 + lane];
                acc[i][j][0] += o.x; acc[i][j][1] += o.y; acc[i][j][2] += o.z; acc[i][j][3] += o.w;
            }
    }
    if (n_splits > 1) {  // fp32 partial slab [split][M][N]
        float* slab = ws + (int64_t)split_id * M * N;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int n = n0 + nb + i * 16 + kq * 4;
            if (n >= N) continue;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int64_t m = m0 + mb + j * 16 + l15;
                if (m < M) *reinterpret_cast<float4*>(slab + m * N + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int n = n0 + nb + i * 16 + kq * 4;
        if (n >= N) continue;
        h4 sh = {(_Float16)1.0f, (_Float16)1.0f, (_Float16)1.0f, (_Float16)1.0f};
        if constexpr (WQ == 8) sh = __builtin_bit_cast(h4, *reinterpret_cast<const uint2*>(scale + n));
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int64_t m = m0 + mb + j * 16 + l15;
            if (m >= M) continue;
            store4<EPI>(yv, ldy, m, n, acc[i][j][0] * (float)sh[0], acc[i][j][1] * (float)sh[1], acc[i][j][2] * (float)sh[2],
                        acc[i][j][3] * (float)sh[3]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Skinny path, M <= 16 rows (decode at small batch): HBM-bound -- every weight byte is read exactly once, straight
// into registers (no LDS round trip: nothing is shared between waves), with several 16-byte loads per lane in flight.
//   grid = N / 16 weight-row tiles; block = 4 waves = 4 contiguous K slices of that tile, summed through LDS;
//   a wave-load fetches 16 rows x 64 contiguous bytes; lane (row = lane & 15, kq = lane >> 4) multiplies its 16 bytes
//   (int8: 16 k, two MFMA k-steps; fp16: 8 k; int4: 32 k, four k-steps) against the matching activation fragment
//   (activations are tiny and L1/L2 resident); up to MT = 2 row tiles of 16 activations reuse each weight fragment.
// ---------------------------------------------------------------------------------------------------------------

}  // namespace pplhip
