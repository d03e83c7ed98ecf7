// K6 / K7 MultiHeadCacheAttention, prefill and cache-prefill (prefix-cache hit) phases: the new tokens of a
// request attend causally to cache positions [0, start_pos + i].  K and V are read back from the KV slab
// (after K5 wrote them), so a cold prefill, a prefix-cache hit (start_pos > 0, ENGINE_CONF_CACHE_PREFILL,
// src/engine/llm_engine.cc:114) and a 1-token request all run this one kernel.  MFMA-bound.
//
// Work decomposition (wave64, gfx950, mfma_f32_16x16x32_f16):
//   grid  = (ceil(max_seq_len / 128), requests, H); block = 8 waves; wave w owns 16 query rows.
//   per KV tile of 128 keys:  K tile -> LDS as fp16 [key][D] (16-B chunks XOR-swizzled against bank conflicts),
//                             V tile -> LDS row-major [16 keys][16 channels] sub-tiles, read TRANSPOSED (ds_read_b64_tr_b16) because the PV
//                             MFMA contracts over keys and needs them contiguous per lane.
//   The tile is staged global -> registers -> LDS; the loads of tile t+1 are issued BEFORE the MFMAs of tile t and
//   converted / written after them (register prefetch, T14 of the CDNA guide), so HBM/L2 latency hides under compute.
//   S^T = K . Q^T  (A = K fragment from LDS, B = Q fragment in registers): the C layout then gives every lane
//   32 scores of ONE query row (col = lane&15), so the online softmax is lane-local plus two xor-shuffles and
//   the probabilities are already in MFMA A-operand order for O += P . V (the k-slot permutation this implies
//   is applied identically to the V^T reads).
//   int8 KV is dequantised to fp16 while staging (one fp16 rounding of q*scale; DESIGN.md "numerics").
// Oracle: ref_attention (oracle/llama_ref.c).
#include <stdlib.h>
#include <type_traits>
#include "kernels.h"

namespace pplhip {

#ifndef PF_NW
#define PF_NW 8   // waves per block, 16 query rows each (4, two blocks per CU: equal at 8192 tokens, 12-18 % slower on shorter prompts -- every staged tile serves half the rows)
#endif
constexpr int PF_BM = 16 * PF_NW;  // query rows per block
#ifndef PF_ABLATE
#define PF_ABLATE 0   // diagnosis builds only (wrong results): 1 no exp, 2 no staging inside the loop, 4 no barriers, 8 no lo MFMA
#endif
constexpr int PF_BN = 128;  // keys per tile
constexpr int PF_VSUB = 272;  // halfs per [16 keys][16 channels] V sub-tile in LDS: 256 + 16 of skew (bank spread of the writes)
constexpr int PF_THREADS = 64 * PF_NW;

template <int D>
__device__ __forceinline__ int k_swz(int key) {
    constexpr int CPR = D / 8;                         // 16-B chunks per row
    constexpr int RPW = (128 / D) > 0 ? (128 / D) : 1; // rows per 256-B bank window
    return (key / RPW) % CPR;
}
// V tile in LDS: row-major [16 keys][16 channels] fp16 sub-tiles, sub-tile (kt, dt) at (kt * D/16 + dt) * PF_VSUB halfs.  The P.V
// MFMA contracts over keys, so its B operand wants 4 keys of ONE channel per lane: gfx950's transposing LDS read delivers exactly
// that from the row-major image (lane l of a 16-lane group supplies the address of row l/4, columns (l%4)*4.. and receives column l
// of the [4][16] block -- profiles/probes/lds_tr_read_probe.hip), so the staging writes V like K (16-byte stores, no shuffling).
typedef short pf_s4 __attribute__((__vector_size__(4 * sizeof(short))));
__device__ __forceinline__ uint2 v_frag_tr(const uint16_t* vs, int sub, int kq, int l15) {
    const uint16_t* p = vs + sub * PF_VSUB + (kq * 4 + (l15 >> 2)) * 16 + (l15 & 3) * 4;
    const pf_s4 w = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) pf_s4*)p);
    return __builtin_bit_cast(uint2, w);
}

// MODE = cache_mode (0 contiguous slots, 1 paged): a compile-time split keeps the page-table load and its wait out of the
// contiguous kernel's prefetch pipeline.  RG = groups of 16 query rows per wave (block = 128 * RG rows): with RG = 2 every staged
// K/V tile (dequantisation + LDS writes by all 512 threads) and every K / V^T fragment read feeds twice the MFMAs -- used for
// long prompts, where the staging is what bounds the kernel.
template <int QBIT, int D, int MODE, int RG>
__global__ __launch_bounds__(PF_THREADS) void attn_prefill_kernel(const uint16_t* __restrict__ qkv, KvAddr kv,
                                                                  const int64_t* __restrict__ seq_starts,
                                                                  const int64_t* __restrict__ start_pos,
                                                                  const int64_t* __restrict__ cache_indices,
                                                                  int64_t max_pages, int64_t b0, int H, int Hkv,
                                                                  int nreq, int nqb, uint16_t* __restrict__ out) {
    constexpr int ELT = QBIT == 8 ? 1 : 2;
    constexpr int CH = 16 / ELT;           // channels in one 16-byte piece
    constexpr int LPT = D / CH;            // pieces per row
    constexpr int KSTEPS = D / 32;
    constexpr int DT = D / 16;
    constexpr int NITEMS = (PF_BN / 2) * LPT;                                // (key pair, piece) staging items per tile
    constexpr int IPT = (NITEMS + PF_THREADS - 1) / PF_THREADS;              // items per thread (1 or 2)
    __shared__ __attribute__((aligned(16))) uint16_t Ks[PF_BN * D];
    __shared__ __attribute__((aligned(16))) uint16_t Vs[(PF_BN / 16) * (D / 16) * PF_VSUB];
    // RG = 2: the Q fragments live in LDS (64 KiB, same chunk swizzle as K) instead of 32 more VGPRs per lane
    __shared__ __attribute__((aligned(16))) uint16_t Qs[RG > 1 ? PF_BM * RG * D : 8];

    // 1-D grid in the XCD-aware order of k_attn_prefill32.hip: XCD id % 8 walks H / 8 consecutive heads one after the other, each head's
    // query tiles heaviest (last) first, so one head's K / V stays in that XCD's L2 and the launch ends with light blocks
    int hq, qb, r;
    {
        const int L = blockIdx.x, per = nqb * nreq;
        int rem;
        if ((H & 7) == 0) {
            const int j = L >> 3;
            hq = (L & 7) * (H >> 3) + j / per;
            rem = j % per;
        } else {
            hq = L / per;
            rem = L % per;
        }
        qb = rem / nreq;
        r = rem % nreq;
    }
    const int64_t b = b0 + r;
    const int hk = hq / (H / Hkv);
    const int64_t seqlen = seq_starts[b + 1] - seq_starts[b];
    constexpr int BM = PF_BM * RG;
    const int64_t q0 = (int64_t)(nqb - 1 - qb) * BM;
    if (q0 >= seqlen) return;
    const int64_t sp = start_pos[b];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int64_t rowstride = (int64_t)(H + 2 * Hkv) * D;

    // Q fragments: B operand, lane (n = query row l15, kq) holds Q[row][ks*32 + kq*8 .. +8]
    const int64_t wrow0 = q0 + wave * (16 * RG);  // first query row of this wave; group g covers wrow0 + 16 g .. + 16
    int64_t qpos[RG];
    h8 qf[RG > 1 ? 1 : RG][KSTEPS];
    f4 o[RG][DT];
    float m[RG], l[RG];
#pragma unroll
    for (int g = 0; g < RG; ++g) {
        int64_t qi = wrow0 + g * 16 + l15;
        if (qi >= seqlen) qi = seqlen - 1;
        qpos[g] = sp + qi;
        const uint16_t* qrow = qkv + (seq_starts[b] + qi) * rowstride + (int64_t)hq * D;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const uint4 v = *reinterpret_cast<const uint4*>(qrow + ks * 32 + kq * 8);
            if constexpr (RG > 1) {
                const int row = wave * (16 * RG) + g * 16 + l15;  // only this wave reads these rows back: no barrier needed
                *reinterpret_cast<uint4*>(&Qs[row * D + ((ks * 4 + kq) ^ k_swz<D>(row)) * 8]) = v;
            } else {
                qf[g][ks] = __builtin_bit_cast(h8, v);
            }
        }
#pragma unroll
        for (int i = 0; i < DT; ++i) o[g][i] = f4{0.f, 0.f, 0.f, 0.f};
        m[g] = -1e30f;
        l[g] = 0.f;
    }
    const float sm_scale2 = 1.4426950408889634f / sqrtf((float)D);  // softmax scale x log2(e)

    const int64_t last_q = (q0 + BM - 1 < seqlen - 1) ? q0 + BM - 1 : seqlen - 1;
    const int64_t kv_end = sp + last_q + 1;   // keys needed by this block: [0, kv_end)
    const int ntiles = (int)((kv_end + PF_BN - 1) / PF_BN);
    // waves whose 16 rows lie entirely beyond the sequence still help staging but skip the MFMAs
    const bool wave_active = wrow0 < seqlen;

    const int64_t slot0 = MODE == 0 ? cache_indices[b] : 0;  // contiguous mode: first slot of the request
    const char* kbase = reinterpret_cast<const char*>(kv.cache) + (int64_t)hk * kv.sH * ELT;
    const char* vbase = kbase + kv.sKV * ELT;
    const uint16_t* ksbase = kv.scale + (int64_t)hk * kv.ssH;
    const uint16_t* vsbase = ksbase + kv.ssKV;

    // ---- register staging of one tile: raw 16-byte pieces of two adjacent keys (+ their int8 group scales) ---------
    uint4 kraw[IPT][2], vraw[IPT][2];
    uint32_t ksc[IPT][2], vsc[IPT][2];  // int8: two fp16 scales (the piece's two groups of 8 channels)
    // Addressing (round 3: the address arithmetic of this lambda was 235 of a tile's ~620 VALU instructions per wave, in a VALU-bound
    // kernel): with contiguous slots the rows of a tile are consecutive slots, so the row base is ONE scalar 64-bit computation
    // per tile and a lane adds a 32-bit offset (key-in-tile x row pitch + piece); with pages, the two keys of an item share a page
    // whenever the page size is even (one table lookup per pair; shift addressing for power-of-two pages).
    const int64_t rowb = kv.sN * ELT, srow = kv.ssN;   // row pitch of the cache (bytes) and of the scales (halfs)
    const int rowb32 = (int)rowb, srow32 = (int)srow;  // a tile spans 128 rows: the lane part fits 32 bits
    const bool pair_in_page = MODE == 1 && (kv.page_size % 2 == 0);
    auto load_tile = [&](int tile) {
        const int64_t key0 = (int64_t)tile * PF_BN;
        const int last = (int)(kv_end - 1 - key0);  // >= 0: keys past kv_end re-read the last valid row (masked later: beyond every row's causal horizon)
#pragma unroll
        for (int it = 0; it < IPT; ++it) {
            const int item = threadIdx.x + it * PF_THREADS;
            if (NITEMS % PF_THREADS == 0 || item < NITEMS) {  // compile-time true for D = 128: no exec branch around the loads
                const int c = item % LPT, kp = item / LPT;
                if constexpr (MODE == 0) {
                    const char* kt = kbase + (slot0 + key0) * rowb;  // tile-uniform
                    const char* vt = vbase + (slot0 + key0) * rowb;
                    const uint16_t* kst = ksbase + (slot0 + key0) * srow;
                    const uint16_t* vst = vsbase + (slot0 + key0) * srow;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int kk = (2 * kp + e) < last ? (2 * kp + e) : last;
                        kraw[it][e] = *reinterpret_cast<const uint4*>(kt + (kk * rowb32 + c * 16));
                        vraw[it][e] = *reinterpret_cast<const uint4*>(vt + (kk * rowb32 + c * 16));
                        if constexpr (QBIT == 8) {
                            ksc[it][e] = *reinterpret_cast<const uint32_t*>(kst + (kk * srow32 + c * 2));
                            vsc[it][e] = *reinterpret_cast<const uint32_t*>(vst + (kk * srow32 + c * 2));
                        }
                    }
                } else {
                    const int k0i = (2 * kp) < last ? (2 * kp) : last, k1i = (2 * kp + 1) < last ? (2 * kp + 1) : last;
                    const int64_t s0 = kv_slot(kv, cache_indices, max_pages, b, key0 + k0i);
                    const int64_t s1 = pair_in_page ? s0 + (k1i - k0i) : kv_slot(kv, cache_indices, max_pages, b, key0 + k1i);
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int64_t slot = e ? s1 : s0;
                        kraw[it][e] = *reinterpret_cast<const uint4*>(kbase + slot * rowb + c * 16);
                        vraw[it][e] = *reinterpret_cast<const uint4*>(vbase + slot * rowb + c * 16);
                        if constexpr (QBIT == 8) {
                            ksc[it][e] = *reinterpret_cast<const uint32_t*>(ksbase + slot * srow + c * 2);
                            vsc[it][e] = *reinterpret_cast<const uint32_t*>(vsbase + slot * srow + c * 2);
                        }
                    }
                }
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int it = 0; it < IPT; ++it) {
            const int item = threadIdx.x + it * PF_THREADS;
            if (NITEMS % PF_THREADS == 0 || item < NITEMS) {  // compile-time true for D = 128: no exec branch around the loads
                const int c = item % LPT, kp = item / LPT;
                const int ch0 = c * CH;
                // the piece as CH/8 groups of 8 fp16, per key of the pair: int8 -> fp16 exactly (v_perm under the exponent),
                // times the group's fp16 scale in packed fp16 (one rounding of q * scale, as the oracle's dequantisation)
                h8 kh[2][CH / 8], vh[2][CH / 8];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    if constexpr (QBIT == 8) {
                        const h8 k0 = cvt_i8x8_f16(make_uint2(kraw[it][e].x, kraw[it][e].y)), k1 = cvt_i8x8_f16(make_uint2(kraw[it][e].z, kraw[it][e].w));
                        const h8 v0 = cvt_i8x8_f16(make_uint2(vraw[it][e].x, vraw[it][e].y)), v1 = cvt_i8x8_f16(make_uint2(vraw[it][e].z, vraw[it][e].w));
                        const h2 ksc2 = __builtin_bit_cast(h2, ksc[it][e]), vsc2 = __builtin_bit_cast(h2, vsc[it][e]);
                        kh[e][0] = k0 * ksc2[0]; kh[e][1] = k1 * ksc2[1];
                        vh[e][0] = v0 * vsc2[0]; vh[e][1] = v1 * vsc2[1];
                    } else {
                        kh[e][0] = __builtin_bit_cast(h8, kraw[it][e]);
                        vh[e][0] = __builtin_bit_cast(h8, vraw[it][e]);
                    }
                }
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int key = 2 * kp + e;
#pragma unroll
                    for (int cc = 0; cc < CH / 8; ++cc) {
                        const int chunk = (ch0 / 8 + cc) ^ k_swz<D>(key);
                        *reinterpret_cast<uint4*>(&Ks[key * D + chunk * 8]) = __builtin_bit_cast(uint4, kh[e][cc]);
                    }
                }
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int key = 2 * kp + e;
#pragma unroll
                    for (int cc = 0; cc < CH / 8; ++cc) {
                        const int ch = ch0 + cc * 8;
                        *reinterpret_cast<uint4*>(&Vs[((key >> 4) * (D / 16) + (ch >> 4)) * PF_VSUB + (key & 15) * 16 + (ch & 15)]) =
                            __builtin_bit_cast(uint4, vh[e][cc]);
                    }
                }
            }
        }
    };

    load_tile(0);
    for (int tile = 0; tile < ntiles; ++tile) {
        const int64_t key0 = (int64_t)tile * PF_BN;
        if (!(PF_ABLATE & 2) || tile == 0) store_tile();
        if (!(PF_ABLATE & 4)) __syncthreads();
        if (tile + 1 < ntiles && !(PF_ABLATE & 2)) load_tile(tile + 1);  // in flight during the MFMAs below

        // a wave whose rows all end before this tile starts has nothing to add (causal); inside an active wave a row group
        // that lies before the tile is merely masked (alpha = 1, all probabilities 0)
        const int64_t wlast = (wrow0 + 16 * RG - 1 < seqlen - 1) ? wrow0 + 16 * RG - 1 : seqlen - 1;
        if (wave_active && key0 <= sp + wlast) {
            // ---- S^T = K . Q^T : 8 key tiles of 16; every K fragment feeds the RG row groups ------------------
            f4 sacc[RG][8];
#pragma unroll
            for (int g = 0; g < RG; ++g)
#pragma unroll
                for (int j = 0; j < 8; ++j) sacc[g][j] = f4{0.f, 0.f, 0.f, 0.f};
            h8 qk[RG];  // RG = 2: this k-step's Q fragments, fetched from LDS
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int key = j * 16 + l15;
                    const int chunk = (ks * 4 + kq) ^ k_swz<D>(key);
                    const h8 a = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(&Ks[key * D + chunk * 8]));
#pragma unroll
                    for (int g = 0; g < RG; ++g) {
                        if constexpr (RG > 1) {
                            if (j == 0) {
                                const int row = wave * (16 * RG) + g * 16 + l15;
                                qk[g] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(&Qs[row * D + ((ks * 4 + kq) ^ k_swz<D>(row)) * 8]));
                            }
                            sacc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, qk[g], sacc[g][j], 0, 0, 0);
                        } else {
                            sacc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, qf[g][ks], sacc[g][j], 0, 0, 0);
                        }
                    }
                }
            }
            // ---- online softmax for query row l15 of each group; this lane holds keys j*16 + kq*4 + r --------
            // scores are kept in the log2 domain (scale * log2(e) folded into one multiply, v_exp_f32 is 2^x); the causal
            // mask costs two VALU per score and is only applied on tiles that reach the wave's diagonal
            const bool need_mask = key0 + PF_BN - 1 > sp + wrow0;  // wave-uniform: some key of the tile may exceed a row's position
#pragma unroll
            for (int g = 0; g < RG; ++g) {
                float mx = -1e30f;
                if (need_mask) {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int64_t kpos = key0 + j * 16 + kq * 4 + r;
                            const float sv = (kpos <= qpos[g]) ? sacc[g][j][r] * sm_scale2 : -1e30f;
                            sacc[g][j][r] = sv;
                            mx = fmaxf(mx, sv);
                        }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float sv = sacc[g][j][r] * sm_scale2;
                            sacc[g][j][r] = sv;
                            mx = fmaxf(mx, sv);
                        }
                }
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float mnew = fmaxf(m[g], mx);
                const float alpha = __builtin_amdgcn_exp2f(m[g] - mnew);
                m[g] = mnew;
                float rs = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = (PF_ABLATE & 1) ? (sacc[g][j][r] - mnew) * 1e-3f : __builtin_amdgcn_exp2f(sacc[g][j][r] - mnew);  // masked scores: 2^(-1e30 - m) = 0
                        sacc[g][j][r] = e;
                        rs += e;
                    }
                rs += __shfl_xor(rs, 16, 64);
                rs += __shfl_xor(rs, 32, 64);
                l[g] = l[g] * alpha + rs;
                // rescale O only when some row's maximum moved (rare after the first tiles): its C layout has rows
                // (kq*4 + r) -> fetch alpha of those query rows
                if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
                    float ar[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) ar[r] = __shfl(alpha, kq * 4 + r, 64);
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[g][dt][r] *= ar[r];
                }
            }
            // ---- O += P . V : A = P (this lane's 8 keys per k-step: tiles 2s, 2s+1), B = V^T from LDS ---------
            // P enters the MFMA as an exact pair of fp16 numbers, hi = fp16(p) and lo = fp16(p - hi) (22 significant bits, two
            // MFMAs): with P rounded to fp16 alone the output differed from the oracle's by one fp16 ulp on a third of its
            // elements (twice the oracle's own summation-order noise on the HF fixtures, tests/test_gpu_model.py)
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                // hi = p truncated to 11 significant bits (a mask: exactly an fp16 number for p >= 2^-14), lo = p - hi (exact in fp32);
                // both packed to fp16 by v_cvt_pkrtz (hi converts exactly, lo keeps 11 more bits): 3 VALU ops per probability
                h8 pa[RG], pl[RG];
#pragma unroll
                for (int g = 0; g < RG; ++g) {
                    typedef __fp16 pk_h2 __attribute__((ext_vector_type(2)));
                    uint32_t hw[4], lw[4];
#pragma unroll
                    for (int q2 = 0; q2 < 4; ++q2) {  // pairs (r, r+1) of tile 2*s2 (q2 = 0, 1) and of tile 2*s2 + 1 (q2 = 2, 3)
                        const float p0 = sacc[g][2 * s2 + (q2 >> 1)][(q2 & 1) * 2], p1 = sacc[g][2 * s2 + (q2 >> 1)][(q2 & 1) * 2 + 1];
                        const float h0 = __uint_as_float(__float_as_uint(p0) & 0xffffe000u), h1 = __uint_as_float(__float_as_uint(p1) & 0xffffe000u);
                        hw[q2] = __builtin_bit_cast(uint32_t, (pk_h2)__builtin_amdgcn_cvt_pkrtz(h0, h1));
                        lw[q2] = __builtin_bit_cast(uint32_t, (pk_h2)__builtin_amdgcn_cvt_pkrtz(p0 - h0, p1 - h1));
                    }
                    pa[g] = __builtin_bit_cast(h8, make_uint4(hw[0], hw[1], hw[2], hw[3]));
                    pl[g] = __builtin_bit_cast(h8, make_uint4(lw[0], lw[1], lw[2], lw[3]));
                }
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const uint2 lo = v_frag_tr(Vs, (2 * s2) * DT + dt, kq, l15);      // keys 16*(2s) + kq*4 .. +4 of channel dt*16 + l15
                    const uint2 hi = v_frag_tr(Vs, (2 * s2 + 1) * DT + dt, kq, l15);
                    const h8 bv = __builtin_bit_cast(h8, make_uint4(lo.x, lo.y, hi.x, hi.y));
#pragma unroll
                    for (int g = 0; g < RG; ++g) {
                        if (!(PF_ABLATE & 8)) o[g][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pl[g], bv, o[g][dt], 0, 0, 0);
                        o[g][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pa[g], bv, o[g][dt], 0, 0, 0);
                    }
                }
            }
        }
        if (!(PF_ABLATE & 4)) __syncthreads();
    }
    // ---- epilogue: O / l, fp16, rows kq*4 + r of every group of this wave ------------------------------------------
#pragma unroll
    for (int g = 0; g < RG; ++g) {
        float lr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) lr[r] = __shfl(l[g], kq * 4 + r, 64);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t qrow_i = wrow0 + g * 16 + kq * 4 + r;
            if (qrow_i < seqlen) {
                uint16_t* orow = out + ((seq_starts[b] + qrow_i) * H + hq) * (int64_t)D;
                const float inv = 1.0f / lr[r];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) orow[dt * 16 + l15] = f2h(o[g][dt][r] * inv);
            }
        }
    }
}

hipError_t launch_attn_prefill(hipStream_t s, const uint16_t* qkv, const KvAddr& kv, int quant_bit,
                               const int64_t* seq_starts, const int64_t* start_pos, const int64_t* cache_indices,
                               int64_t max_pages, int64_t b0, int64_t B, int H, int Hkv, int D, int64_t max_seq_len,
                               uint16_t* out, int64_t max_kv_len, float* ws, size_t ws_bytes, int64_t row0, int64_t nrows) {
    if (B <= b0 || max_seq_len <= 0) return hipSuccess;
    // head_dim 128: the 32-row kernel of k_attn_prefill32.hip (PPLHIP_PREFILL32=0 selects this file's 16-row kernel for A/B runs)
    static const int p32 = tune_int("PPLHIP_PREFILL32", 1);
    if (p32 && D == 128 && (quant_bit == 0 || quant_bit == 8)) return launch_attn_prefill32(s, qkv, kv, quant_bit, seq_starts, start_pos, cache_indices, max_pages, b0, B, H, Hkv, D, max_seq_len, out, max_kv_len, ws, ws_bytes, row0, nrows);
    // RG = 2 (256 query rows per block, Q fragments in LDS) halves the staging per MFMA but spills registers and measured
    // slower than RG = 1 once the softmax was trimmed (8192-token prompt: 1.86 ms vs 1.43 ms per layer); PPLHIP_PREFILL_RG=2 keeps
    // it reachable for experiments
    static const int forced_rg = tune_int("PPLHIP_PREFILL_RG", 0);
    const int rg = forced_rg == 2 ? 2 : 1;
    const int bm = PF_BM * rg;
    const int nqb = (int)((max_seq_len + bm - 1) / bm), nreq = (int)(B - b0);
    dim3 grid((unsigned)((int64_t)nqb * nreq * H));
#define PF_LAUNCH(QB, DD, MD, RGV)                                                                                     \
    hipLaunchKernelGGL((attn_prefill_kernel<QB, DD, MD, RGV>), grid, dim3(PF_THREADS), 0, s, qkv, kv, seq_starts, start_pos, \
                       cache_indices, max_pages, b0, H, Hkv, nreq, nqb, out)
#define PF_CASE(QB, DD)                                                                                          \
    if (quant_bit == QB && D == DD) {                                                                            \
        if (kv.mode == 0) { if (rg == 2) PF_LAUNCH(QB, DD, 0, 2); else PF_LAUNCH(QB, DD, 0, 1); }                \
        else { if (rg == 2) PF_LAUNCH(QB, DD, 1, 2); else PF_LAUNCH(QB, DD, 1, 1); }                             \
        return hipGetLastError();                                                                                \
    }
    PF_CASE(8, 128) PF_CASE(0, 128) PF_CASE(8, 64) PF_CASE(0, 64) PF_CASE(8, 32) PF_CASE(0, 32)
#undef PF_LAUNCH
#undef PF_CASE
    return hipErrorInvalidValue;
}

}  // namespace pplhip
