// K6 / K7 prefill and cache-prefill attention, head_dim 128 -- the 32-row wave tile (round 3).
//
// k_attn_prefill.hip's kernel (8 waves x 16 query rows, mfma_f32_16x16x32_f16, 128-key tiles) is bound by its own read / MFMA /
// softmax core at 14-20 % of the MFMA peak (profiles/r03_prefill_attention_ablation.md).  This one follows the structure the CDNA4
// guide's attention ladder ends on: a wave owns 32 query rows and works on mfma_f32_32x32x16_f16 -- per (query, key) pair half the LDS
// fragment traffic and half the MFMA / read instructions of the 16-row form --, 64-key tiles double-buffered in LDS with ONE
// barrier per tile, the next tile's global loads issued before the tile's MFMAs and dequantised / written after them.
//   grid = (ceil(max_seq_len / 128), requests, H); block = 4 waves x 32 query rows.
//   S^T = K . Q^T (A = K fragment from LDS, B = Q fragment in registers): lane (query l31 = lane & 31, half hi = lane >> 5) then holds
//   16 scores of ITS query per 32-key block, keys crow(i, hi) = (i & 3) + 8 (i >> 2) + 4 hi -- the softmax is lane-local plus one
//   exchange with lane ^ 32, and the probabilities are already the A operand of O += P . V: k-step s of P . V (keys 16 s .. + 16)
//   takes the lane's values i = 8 (s & 1) .. + 8 of block s >> 1, i.e. keys {4 hi .. + 4, 8 + 4 hi .. + 4} + 16 s, and the V^T fragment
//   is read with the same key permutation (two transposing LDS reads of 4 keys each).
//   P enters as an exact hi + lo pair of fp16 numbers (two MFMAs), K / V dequantised int8 x scale rounded to fp16 once, as in the
//   16-row kernel (DESIGN.md numerics).
// Oracle: ref_attention (oracle/llama_ref.c).
#include <stdlib.h>
#include "kernels.h"

namespace pplhip {

namespace {

constexpr int P3_BM = 128, P3_BN = 64, P3_THREADS = 256, P3_D = 128;
constexpr int P3_VSUB = 272;  // halfs per [16 keys][16 channels] V sub-tile: 256 + 16 of skew
constexpr int P3_KS_HALFS = P3_BN * P3_D, P3_VS_HALFS = (P3_BN / 16) * (P3_D / 16) * P3_VSUB;
typedef float f16v __attribute__((ext_vector_type(16)));
typedef short p3_s4 __attribute__((__vector_size__(4 * sizeof(short))));

__device__ __forceinline__ uint2 p3_v_frag(const uint16_t* sub, int r0, int l15) {
    const uint16_t* p = sub + (r0 + (l15 >> 2)) * 16 + (l15 & 3) * 4;
    const p3_s4 w = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) p3_s4*)p);
    return __builtin_bit_cast(uint2, w);
}
__device__ __forceinline__ int crow(int i, int hi) { return (i & 3) + 8 * (i >> 2) + 4 * hi; }

template <int QBIT, int MODE>
__global__ __launch_bounds__(P3_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_prefill32_kernel(const uint16_t* __restrict__ qkv, KvAddr kv,
                                                                    const int64_t* __restrict__ seq_starts,
                                                                    const int64_t* __restrict__ start_pos,
                                                                    const int64_t* __restrict__ cache_indices, int64_t max_pages,
                                                                    int64_t b0, int H, int Hkv, uint16_t* __restrict__ out) {
    constexpr int D = P3_D;
    constexpr int ELT = QBIT == 8 ? 1 : 2;
    constexpr int CH = 16 / ELT;                 // channels in one 16-byte piece
    constexpr int LPT = D / CH;                  // pieces per row: 8 (int8) / 16 (fp16)
    constexpr int IPT = P3_BN * LPT / P3_THREADS;  // (key, piece) items per thread and matrix: 2 / 4
    constexpr int KSTEPS = D / 16;               // 8 k-steps of the 32x32x16 MFMA over the head dimension
    __shared__ __attribute__((aligned(16))) uint16_t smem[2 * (P3_KS_HALFS + P3_VS_HALFS)];

    const int64_t b = b0 + blockIdx.y;
    const int hq = blockIdx.z;
    const int hk = hq / (H / Hkv);
    const int64_t seqlen = seq_starts[b + 1] - seq_starts[b];
    const int64_t q0 = (int64_t)(gridDim.x - 1 - blockIdx.x) * P3_BM;  // heaviest (last) query tiles first
    if (q0 >= seqlen) return;
    const int64_t sp = start_pos[b];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, hi = lane >> 5, l15 = lane & 15, chblk = (lane >> 4) & 1;
    const int64_t rowstride = (int64_t)(H + 2 * Hkv) * D;

    const int64_t wrow0 = q0 + wave * 32;
    int64_t qi = wrow0 + l31;
    if (qi >= seqlen) qi = seqlen - 1;
    const int64_t qpos = sp + qi;
    h8 qf[KSTEPS];  // B operand: lane (query l31, hi) holds channels 16 ks + 8 hi .. + 8
    {
        const uint16_t* qrow = qkv + (seq_starts[b] + qi) * rowstride + (int64_t)hq * D;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) qf[ks] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(qrow + ks * 16 + hi * 8));
    }
    f16v o[4];  // O^T blocks: lane = channel c32 * 32 + l31, values i = queries crow(i, hi)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[c][i] = 0.f;
    float m = -1e30f, l = 0.f;
    const float sm_scale2 = 1.4426950408889634f / sqrtf((float)D);  // softmax scale x log2(e)

    const int64_t last_q = (q0 + P3_BM - 1 < seqlen - 1) ? q0 + P3_BM - 1 : seqlen - 1;
    const int64_t kv_end = sp + last_q + 1;
    const int ntiles = (int)((kv_end + P3_BN - 1) / P3_BN);
    const bool wave_active = wrow0 < seqlen;
    const int64_t wlast = (wrow0 + 31 < seqlen - 1) ? wrow0 + 31 : seqlen - 1;

    const int64_t slot0 = MODE == 0 ? cache_indices[b] : 0;
    const char* kbase = reinterpret_cast<const char*>(kv.cache) + (int64_t)hk * kv.sH * ELT;
    const char* vbase = kbase + kv.sKV * ELT;
    const uint16_t* ksbase = kv.scale + (int64_t)hk * kv.ssH;
    const uint16_t* vsbase = ksbase + kv.ssKV;
    const int64_t rowb = kv.sN * ELT, srow = kv.ssN;
    const int rowb32 = (int)rowb, srow32 = (int)srow;

    // ---- staging registers of one 64-key tile: item (key, piece) = divmod(tid + 256 it, LPT).  (Macros, not lambdas: register arrays
    // captured by reference end up in scratch memory -- profiles/r02_gemm_experiments.md.)
    uint4 kraw[IPT], vraw[IPT];
    uint32_t ksc[IPT], vsc[IPT];
#define P3_ITEM_KEY(it) ((int)(threadIdx.x + (it) * P3_THREADS) / LPT)
#define P3_ITEM_PC(it) ((int)(threadIdx.x + (it) * P3_THREADS) % LPT)
    // rows past kv_end re-read the last valid row (beyond every row's causal horizon); contiguous slots: one scalar tile base plus
    // 32-bit lane offsets
#define P3_LOAD_TILE(TILE)                                                                                                     \
    do {                                                                                                                       \
        const int64_t key0_ = (int64_t)(TILE) * P3_BN;                                                                         \
        const int last_ = (int)(kv_end - 1 - key0_);                                                                           \
        _Pragma("unroll") for (int it = 0; it < IPT; ++it) {                                                                   \
            const int kk_ = P3_ITEM_KEY(it) < last_ ? P3_ITEM_KEY(it) : last_, pc_ = P3_ITEM_PC(it);                           \
            const char *kp_, *vp_;                                                                                             \
            const uint16_t *ksp_, *vsp_;                                                                                       \
            if constexpr (MODE == 0) {                                                                                         \
                const int64_t sb_ = slot0 + key0_;                                                                             \
                kp_ = kbase + sb_ * rowb + (kk_ * rowb32 + pc_ * 16);                                                          \
                vp_ = vbase + sb_ * rowb + (kk_ * rowb32 + pc_ * 16);                                                          \
                ksp_ = ksbase + sb_ * srow + (kk_ * srow32 + pc_ * 2);                                                         \
                vsp_ = vsbase + sb_ * srow + (kk_ * srow32 + pc_ * 2);                                                         \
            } else {                                                                                                           \
                const int64_t slot_ = kv_slot(kv, cache_indices, max_pages, b, key0_ + kk_);                                   \
                kp_ = kbase + slot_ * rowb + pc_ * 16;                                                                         \
                vp_ = vbase + slot_ * rowb + pc_ * 16;                                                                         \
                ksp_ = ksbase + slot_ * srow + pc_ * 2;                                                                        \
                vsp_ = vsbase + slot_ * srow + pc_ * 2;                                                                        \
            }                                                                                                                  \
            kraw[it] = *reinterpret_cast<const uint4*>(kp_);                                                                   \
            vraw[it] = *reinterpret_cast<const uint4*>(vp_);                                                                   \
            if constexpr (QBIT == 8) {                                                                                         \
                ksc[it] = *reinterpret_cast<const uint32_t*>(ksp_);                                                            \
                vsc[it] = *reinterpret_cast<const uint32_t*>(vsp_);                                                            \
            }                                                                                                                  \
        }                                                                                                                      \
    } while (0)

    P3_LOAD_TILE(0);
    for (int tile = 0; tile < ntiles; ++tile) {
        const int64_t key0 = (int64_t)tile * P3_BN;
        {   // dequantise the tile loaded during the previous iteration -> stage tile & 1.  K: fp16 [64 keys][128 channels], 16-byte
            // chunks XOR-swizzled with (key & 15) (conflict-free b128 fragment reads); V: row-major [16 keys][16 channels] sub-tiles
            uint16_t* Kw = smem + (tile & 1) * (P3_KS_HALFS + P3_VS_HALFS);
            uint16_t* Vw = Kw + P3_KS_HALFS;
#pragma unroll
            for (int it = 0; it < IPT; ++it) {
                const int key = P3_ITEM_KEY(it), ch0 = P3_ITEM_PC(it) * CH;
                if constexpr (QBIT == 8) {
                    const h8 k0 = cvt_i8x8_f16(make_uint2(kraw[it].x, kraw[it].y)), k1 = cvt_i8x8_f16(make_uint2(kraw[it].z, kraw[it].w));
                    const h8 v0 = cvt_i8x8_f16(make_uint2(vraw[it].x, vraw[it].y)), v1 = cvt_i8x8_f16(make_uint2(vraw[it].z, vraw[it].w));
                    const h2 ks2 = __builtin_bit_cast(h2, ksc[it]), vs2 = __builtin_bit_cast(h2, vsc[it]);
                    const h8 kh0 = k0 * ks2[0], kh1 = k1 * ks2[1], vh0 = v0 * vs2[0], vh1 = v1 * vs2[1];  // one rounding of q * scale
                    *reinterpret_cast<uint4*>(&Kw[key * D + ((ch0 >> 3) ^ (key & 15)) * 8]) = __builtin_bit_cast(uint4, kh0);
                    *reinterpret_cast<uint4*>(&Kw[key * D + (((ch0 >> 3) + 1) ^ (key & 15)) * 8]) = __builtin_bit_cast(uint4, kh1);
                    uint16_t* vd = &Vw[((key >> 4) * (D / 16) + (ch0 >> 4)) * P3_VSUB + (key & 15) * 16];
                    *reinterpret_cast<uint4*>(vd) = __builtin_bit_cast(uint4, vh0);
                    *reinterpret_cast<uint4*>(vd + 8) = __builtin_bit_cast(uint4, vh1);
                } else {
                    *reinterpret_cast<uint4*>(&Kw[key * D + ((ch0 >> 3) ^ (key & 15)) * 8]) = kraw[it];
                    *reinterpret_cast<uint4*>(&Vw[((key >> 4) * (D / 16) + (ch0 >> 4)) * P3_VSUB + (key & 15) * 16 + (ch0 & 15)]) = vraw[it];
                }
            }
        }
        __syncthreads();  // tile `tile` is visible in stage tile & 1; every wave is past its reads of the other stage (tile - 1)
        // in flight during the MFMAs below.  UNCONDITIONAL (the last iteration re-reads its own tile and drops it): a conditional
        // refill made hipcc keep the fp16 build's staging arrays in scratch memory
        P3_LOAD_TILE(tile + 1 < ntiles ? tile + 1 : tile);
        const uint16_t* Ks = smem + (tile & 1) * (P3_KS_HALFS + P3_VS_HALFS);
        const uint16_t* Vs = Ks + P3_KS_HALFS;
        if (wave_active && key0 <= sp + wlast) {  // causal: a wave whose rows all end before this tile has nothing to add
            // ---- S^T = K . Q^T: two 32-key blocks ----------------------------------------------------------------------------
            f16v sacc[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) sacc[kb][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const int key = kb * 32 + l31;
                    const h8 a = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(&Ks[key * D + ((2 * ks + hi) ^ (key & 15)) * 8]));
                    sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qf[ks], sacc[kb], 0, 0, 0);
                }
            // ---- online softmax of query l31 (log2 domain); this lane holds keys key0 + 32 kb + crow(i, hi) ------------------------
            const bool need_mask = key0 + P3_BN - 1 > sp + wrow0;  // wave-uniform: some key of the tile may exceed a row's position
            float mx = -1e30f;
            if (need_mask) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int64_t kpos = key0 + 32 * kb + crow(i, hi);
                        const float sv = (kpos <= qpos) ? sacc[kb][i] * sm_scale2 : -1e30f;
                        sacc[kb][i] = sv;
                        mx = fmaxf(mx, sv);
                    }
            } else {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float sv = sacc[kb][i] * sm_scale2;
                        sacc[kb][i] = sv;
                        mx = fmaxf(mx, sv);
                    }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mnew = fmaxf(m, mx);
            const float alpha = __builtin_amdgcn_exp2f(m - mnew);
            m = mnew;
            float rs = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float e = __builtin_amdgcn_exp2f(sacc[kb][i] - mnew);  // masked scores: 2^(-1e30 - m) = 0
                    sacc[kb][i] = e;
                    rs += e;
                }
            rs += __shfl_xor(rs, 32, 64);
            l = l * alpha + rs;
            // rescale O only when some row's maximum moved (rare after the first tiles): O's values are queries crow(i, hi)
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float ar = __shfl(alpha, crow(i, hi), 64);
#pragma unroll
                    for (int c = 0; c < 4; ++c) o[c][i] *= ar;
                }
            }
            // ---- O += P . V over four 16-key k-steps; P as an exact hi + lo pair (mask, subtract, v_cvt_pkrtz) ---------------------
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                typedef __fp16 pk_h2 __attribute__((ext_vector_type(2)));
                uint32_t hw[4], lw[4];
#pragma unroll
                for (int q2 = 0; q2 < 4; ++q2) {
                    const float p0 = sacc[s >> 1][8 * (s & 1) + 2 * q2], p1 = sacc[s >> 1][8 * (s & 1) + 2 * q2 + 1];
                    const float h0 = __uint_as_float(__float_as_uint(p0) & 0xffffe000u), h1 = __uint_as_float(__float_as_uint(p1) & 0xffffe000u);
                    hw[q2] = __builtin_bit_cast(uint32_t, (pk_h2)__builtin_amdgcn_cvt_pkrtz(h0, h1));
                    lw[q2] = __builtin_bit_cast(uint32_t, (pk_h2)__builtin_amdgcn_cvt_pkrtz(p0 - h0, p1 - h1));
                }
                const h8 pa = __builtin_bit_cast(h8, make_uint4(hw[0], hw[1], hw[2], hw[3]));
                const h8 pl = __builtin_bit_cast(h8, make_uint4(lw[0], lw[1], lw[2], lw[3]));
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint16_t* sub = Vs + (s * (D / 16) + 2 * c + chblk) * P3_VSUB;  // keys 16 s .. + 16, channels 32 c + 16 chblk .. + 16
                    const uint2 v0 = p3_v_frag(sub, 4 * hi, l15), v1 = p3_v_frag(sub, 8 + 4 * hi, l15);
                    const h8 bv = __builtin_bit_cast(h8, make_uint4(v0.x, v0.y, v1.x, v1.y));
                    o[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl, bv, o[c], 0, 0, 0);
                    o[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa, bv, o[c], 0, 0, 0);
                }
            }
        }
    }
    // ---- epilogue: O / l, fp16; value i of channel block c = query wrow0 + crow(i, hi), channel 32 c + l31 ----------------------------
    if (!wave_active) return;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = crow(i, hi);
        const float lr = __shfl(l, r, 64);
        const int64_t qrow_i = wrow0 + r;
        if (qrow_i < seqlen) {
            uint16_t* orow = out + ((seq_starts[b] + qrow_i) * H + hq) * (int64_t)D;
            const float inv = 1.0f / lr;
#pragma unroll
            for (int c = 0; c < 4; ++c) orow[c * 32 + l31] = f2h(o[c][i] * inv);
        }
    }
}

#undef P3_LOAD_TILE
#undef P3_ITEM_KEY
#undef P3_ITEM_PC

}  // namespace

hipError_t launch_attn_prefill32(hipStream_t s, const uint16_t* qkv, const KvAddr& kv, int quant_bit, const int64_t* seq_starts,
                                 const int64_t* start_pos, const int64_t* cache_indices, int64_t max_pages, int64_t b0, int64_t B,
                                 int H, int Hkv, int D, int64_t max_seq_len, uint16_t* out) {
    if (D != P3_D || (quant_bit != 0 && quant_bit != 8)) return hipErrorInvalidValue;
    if (B <= b0 || max_seq_len <= 0) return hipSuccess;
    dim3 grid((unsigned)((max_seq_len + P3_BM - 1) / P3_BM), (unsigned)(B - b0), (unsigned)H);
#define P3_LAUNCH(QB, MD) hipLaunchKernelGGL((attn_prefill32_kernel<QB, MD>), grid, dim3(P3_THREADS), 0, s, qkv, kv, seq_starts, start_pos, \
                                             cache_indices, max_pages, b0, H, Hkv, out)
    if (quant_bit == 8) { if (kv.mode == 0) P3_LAUNCH(8, 0); else P3_LAUNCH(8, 1); }
    else { if (kv.mode == 0) P3_LAUNCH(0, 0); else P3_LAUNCH(0, 1); }
#undef P3_LAUNCH
    return hipGetLastError();
}

}  // namespace pplhip
