// K6 / K7 prefill and cache-prefill attention, head_dim 128 -- the 32-row wave tile (round 3).
//
// k_attn_prefill.hip's kernel (8 waves x 16 query rows, mfma_f32_16x16x32_f16, 128-key tiles) is bound by its own read / MFMA /
// softmax core at 14-20 % of the MFMA peak (profiles/r03_prefill_attention_ablation.md).  This one follows the structure the CDNA4
// guide's attention ladder ends on: a wave owns 32 query rows and works on mfma_f32_32x32x16_f16 -- per (query, key) pair half the LDS
// fragment traffic and half the MFMA / read instructions of the 16-row form --, 64-key tiles double-buffered in LDS with ONE
// barrier per tile, the next tile's global loads issued before the tile's MFMAs and dequantised / written after them.
//   grid = ceil(max_seq_len / BM) x requests x H workgroups in an XCD-aware order; block = 4 or 8 waves x 32 query rows (BM = 128 / 256).
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

constexpr int P3_BN = 64, P3_D = 128;
// P enters P . V as an exact hi + lo pair of fp16 numbers (two MFMAs per block): DECIDED in round 5, the build switch of rounds 3-4 is gone.
// P rounded to nearest once measured 8192-token prompt 1002 -> 820 us (549 -> 670 TFLOP/s counted) with every test green, but it raises the
// MODEL-level distance to the oracle where the margin is thinnest -- 32-layer 7B logits 1.14e-2 -> 1.35e-2 (1.03 -> 1.22 x the noise floor,
// bar 1.3), config 5's cache-prefill step 1.05e-3 -> 1.17e-3 (1.5 x its floor); profiles/r05_parity_prefill_p_{exact,rounded}.jsonl -- the same
// trade that the grouped-query decode kernel's rounded V turned out to be (k_attn_decode_gqa.hip).  Precision first: ~5 % of a cold TTFT.
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

// SPLIT: the block handles the KV tiles of split blockIdx.y of gridDim.y only and writes unnormalised partial rows (O, m, l) to `ws`
// ([token row - ws_row0][head][split][D + 2] floats, the layout of the decode kernels: the workspace holds the launch's own rows only,
// whatever decode rows or earlier chunks precede them in the step); attn_prefill32_reduce_kernel merges them.  For short new
// suffixes behind a long cached prefix (a prefix-cache hit recomputes one page): a request is otherwise H blocks walking thousands of keys.
template <int QBIT, int MODE, int NW, int ABL = 0, int SPLIT = 0>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_prefill32_kernel(const uint16_t* __restrict__ qkv, KvAddr kv,
                                                                    const int64_t* __restrict__ seq_starts,
                                                                    const int64_t* __restrict__ start_pos,
                                                                    const int64_t* __restrict__ cache_indices, int64_t max_pages,
                                                                    int64_t b0, int H, int Hkv, int nreq, int nqb, uint16_t* __restrict__ out,
                                                                    float* __restrict__ ws, int64_t ws_row0) {
    constexpr int D = P3_D;
    constexpr int P3_BM = NW * 32, P3_THREADS = NW * 64;  // NW waves x 32 query rows share the staged K / V tiles
    constexpr int ELT = QBIT == 8 ? 1 : 2;
    constexpr int CH = 16 / ELT;                 // channels in one 16-byte piece
    constexpr int LPT = D / CH;                  // pieces per row: 8 (int8) / 16 (fp16)
    constexpr int IPT = P3_BN * LPT / P3_THREADS;  // (key, piece) items per thread and matrix: 2 / 4 (4 waves), 1 / 2 (8 waves)
    constexpr int KSTEPS = D / 16;               // 8 k-steps of the 32x32x16 MFMA over the head dimension
    __shared__ __attribute__((aligned(16))) uint16_t smem[2 * (P3_KS_HALFS + P3_VS_HALFS)];

    // 1-D grid, XCD-aware: consecutive workgroup ids go round-robin over the 8 XCDs (each with its own 4 MiB L2), so XCD x takes the
    // H / 8 consecutive query heads x H/8 .. (same KV head under grouped-query attention) and walks them one after the other, each head's
    // query blocks heaviest (last) first: what an XCD runs at any moment reads ONE head's K / V, which then lives in its L2
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
    const int64_t q0 = (int64_t)(nqb - 1 - qb) * P3_BM;
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
    uint4 kr0, kr1, kr2, kr3, vr0, vr1, vr2, vr3;  // scalars: hipcc kept `uint4 kraw[IPT]` of the fp16 build (IPT = 4) in scratch memory
    uint32_t kc0 = 0, kc1 = 0, kc2 = 0, kc3 = 0, vc0 = 0, vc1 = 0, vc2 = 0, vc3 = 0;
    kr1 = kr2 = kr3 = vr1 = vr2 = vr3 = make_uint4(0, 0, 0, 0);
#define P3_ITEM_KEY(it) ((int)(threadIdx.x + (it) * P3_THREADS) / LPT)
#define P3_ITEM_PC(it) ((int)(threadIdx.x + (it) * P3_THREADS) % LPT)
    // rows past kv_end re-read the last valid row (beyond every row's causal horizon); contiguous slots: one scalar tile base plus
    // 32-bit lane offsets
#define P3_LOAD_ITEM(it, KR, VR, KC, VC)                                                                                        \
    do {                                                                                                                       \
        const int kk_ = P3_ITEM_KEY(it) < last_ ? P3_ITEM_KEY(it) : last_, pc_ = P3_ITEM_PC(it);                               \
        const char *kp_, *vp_;                                                                                                 \
        const uint16_t *ksp_, *vsp_;                                                                                           \
        if constexpr (MODE == 0) {                                                                                             \
            const int64_t sb_ = slot0 + key0_;                                                                                 \
            kp_ = kbase + sb_ * rowb + (kk_ * rowb32 + pc_ * 16);                                                              \
            vp_ = vbase + sb_ * rowb + (kk_ * rowb32 + pc_ * 16);                                                              \
            ksp_ = ksbase + sb_ * srow + (kk_ * srow32 + pc_ * 2);                                                             \
            vsp_ = vsbase + sb_ * srow + (kk_ * srow32 + pc_ * 2);                                                             \
        } else {                                                                                                               \
            const int64_t slot_ = kv_slot(kv, cache_indices, max_pages, b, key0_ + kk_);                                       \
            kp_ = kbase + slot_ * rowb + pc_ * 16;                                                                             \
            vp_ = vbase + slot_ * rowb + pc_ * 16;                                                                             \
            ksp_ = ksbase + slot_ * srow + pc_ * 2;                                                                            \
            vsp_ = vsbase + slot_ * srow + pc_ * 2;                                                                            \
        }                                                                                                                      \
        KR = *reinterpret_cast<const uint4*>(kp_);                                                                             \
        VR = *reinterpret_cast<const uint4*>(vp_);                                                                             \
        if constexpr (QBIT == 8) {                                                                                             \
            KC = *reinterpret_cast<const uint32_t*>(ksp_);                                                                     \
            VC = *reinterpret_cast<const uint32_t*>(vsp_);                                                                     \
        }                                                                                                                      \
    } while (0)
#define P3_LOAD_TILE(TILE)                                                                                                     \
    do {                                                                                                                       \
        const int64_t key0_ = (int64_t)(TILE) * P3_BN;                                                                         \
        const int last_ = (int)(kv_end - 1 - key0_);                                                                           \
        P3_LOAD_ITEM(0, kr0, vr0, kc0, vc0);                                                                                   \
        if constexpr (IPT >= 2) P3_LOAD_ITEM(1, kr1, vr1, kc1, vc1);                                                           \
        if constexpr (IPT == 4) {                                                                                              \
            P3_LOAD_ITEM(2, kr2, vr2, kc2, vc2);                                                                               \
            P3_LOAD_ITEM(3, kr3, vr3, kc3, vc3);                                                                               \
        }                                                                                                                      \
    } while (0)
    // dequantise one item -> stage.  K: fp16 [64 keys][128 channels], 16-byte chunks XOR-swizzled with (key & 15) (conflict-free b128
    // fragment reads); V: row-major [16 keys][16 channels] sub-tiles
#define P3_STORE_ITEM(it, KR, VR, KC, VC)                                                                                       \
    do {                                                                                                                       \
        const int key = P3_ITEM_KEY(it), ch0 = P3_ITEM_PC(it) * CH;                                                            \
        if constexpr (QBIT == 8) {                                                                                             \
            const h8 k0 = cvt_i8x8_f16(make_uint2(KR.x, KR.y)), k1 = cvt_i8x8_f16(make_uint2(KR.z, KR.w));                     \
            const h8 v0 = cvt_i8x8_f16(make_uint2(VR.x, VR.y)), v1 = cvt_i8x8_f16(make_uint2(VR.z, VR.w));                     \
            const h2 ks2 = __builtin_bit_cast(h2, KC), vs2 = __builtin_bit_cast(h2, VC);                                       \
            const h8 kh0 = k0 * ks2[0], kh1 = k1 * ks2[1], vh0 = v0 * vs2[0], vh1 = v1 * vs2[1]; /* one rounding of q x scale */ \
            *reinterpret_cast<uint4*>(&Kw[key * D + ((ch0 >> 3) ^ (key & 15)) * 8]) = __builtin_bit_cast(uint4, kh0);          \
            *reinterpret_cast<uint4*>(&Kw[key * D + (((ch0 >> 3) + 1) ^ (key & 15)) * 8]) = __builtin_bit_cast(uint4, kh1);    \
            uint16_t* vd = &Vw[((key >> 4) * (D / 16) + (ch0 >> 4)) * P3_VSUB + (key & 15) * 16];                              \
            *reinterpret_cast<uint4*>(vd) = __builtin_bit_cast(uint4, vh0);                                                    \
            *reinterpret_cast<uint4*>(vd + 8) = __builtin_bit_cast(uint4, vh1);                                                \
        } else {                                                                                                               \
            *reinterpret_cast<uint4*>(&Kw[key * D + ((ch0 >> 3) ^ (key & 15)) * 8]) = KR;                                      \
            *reinterpret_cast<uint4*>(&Vw[((key >> 4) * (D / 16) + (ch0 >> 4)) * P3_VSUB + (key & 15) * 16 + (ch0 & 15)]) = VR; \
        }                                                                                                                      \
    } while (0)

    // SPLIT: tiles [tb, te) of this split (possibly none)
    const int tb = SPLIT ? (int)((int64_t)ntiles * blockIdx.y / gridDim.y) : 0;
    const int te = SPLIT ? (int)((int64_t)ntiles * (blockIdx.y + 1) / gridDim.y) : ntiles;
    if (tb < te) P3_LOAD_TILE(tb);
    for (int tile = tb; tile < te; ++tile) {
        const int64_t key0 = (int64_t)tile * P3_BN;
        if (!(ABL & 2) || tile < tb + 2) {   // dequantise the tile loaded during the previous iteration -> stage tile & 1
            uint16_t* Kw = smem + ((tile - tb) & 1) * (P3_KS_HALFS + P3_VS_HALFS);
            uint16_t* Vw = Kw + P3_KS_HALFS;
            P3_STORE_ITEM(0, kr0, vr0, kc0, vc0);
            if constexpr (IPT >= 2) P3_STORE_ITEM(1, kr1, vr1, kc1, vc1);
            if constexpr (IPT == 4) {
                P3_STORE_ITEM(2, kr2, vr2, kc2, vc2);
                P3_STORE_ITEM(3, kr3, vr3, kc3, vc3);
            }
        }
        if (!(ABL & 8)) __syncthreads();  // tile `tile` is visible in stage tile & 1; every wave is past its reads of the other stage (tile - 1)
        // in flight during the MFMAs below.  UNCONDITIONAL (the last iteration re-reads its own tile and drops it): a conditional
        // refill made hipcc keep the fp16 build's staging arrays in scratch memory
        if (!(ABL & 2)) P3_LOAD_TILE(tile + 1 < te ? tile + 1 : tile);
        const uint16_t* Ks = smem + ((tile - tb) & 1) * (P3_KS_HALFS + P3_VS_HALFS);
        const uint16_t* Vs = Ks + P3_KS_HALFS;
        if (wave_active && key0 <= sp + wlast) {  // causal: a wave whose rows all end before this tile has nothing to add
            // ---- S^T = K . Q^T: two 32-key blocks ----------------------------------------------------------------------------
            f16v sacc[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) sacc[kb][i] = 0.f;
            // K fragments two k-steps ahead of their MFMAs, in named registers (left to itself hipcc funnels all sixteen reads through
            // one register pair: read, wait, two MFMAs, read ...)
#define P3_KFRAG(ks, kb) __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(&Ks[((kb) * 32 + l31) * D + ((2 * (ks) + hi) ^ (l31 & 15)) * 8]))
            h8 ka0 = P3_KFRAG(0, 0), kb0 = P3_KFRAG(0, 1), ka1 = P3_KFRAG(1, 0), kb1 = P3_KFRAG(1, 1);
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ks += 2) {
                sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka0, qf[ks], sacc[0], 0, 0, 0);
                sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kb0, qf[ks], sacc[1], 0, 0, 0);
                if (ks + 2 < KSTEPS) { ka0 = P3_KFRAG(ks + 2, 0); kb0 = P3_KFRAG(ks + 2, 1); }
                sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka1, qf[ks + 1], sacc[0], 0, 0, 0);
                sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kb1, qf[ks + 1], sacc[1], 0, 0, 0);
                if (ks + 3 < KSTEPS) { ka1 = P3_KFRAG(ks + 3, 0); kb1 = P3_KFRAG(ks + 3, 1); }
            }
#undef P3_KFRAG
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);  // pin that order: 4 reads | (2 MFMAs, 2 reads) x 6 | 4 MFMAs
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                if (ks + 2 < KSTEPS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
            // ---- online softmax of query l31; this lane holds keys key0 + 32 kb + crow(i, hi).  The running maximum m stays in
            // RAW score units; scale x log2(e) is folded into the exponent's fma: p = 2^(s c - m c).  Masked scores are -inf while m
            // starts FINITE (-1e30): a row whose first tile of a split is wholly masked keeps m, gets alpha = 2^0 and p = 2^(-inf) = 0
            // exactly, whatever the rounding of m c (with a -1e30 mask p was 2^(rounding residual)) ------------------------
            if (key0 + P3_BN - 1 > sp + wrow0) {  // wave-uniform: some key of the tile may exceed a row's position
                const int64_t d = qpos - key0;
                const int qrel = d < 0 ? -1 : (d > 2 * P3_BN ? 2 * P3_BN : (int)d);  // keys of the tile with index <= qrel are visible
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        sacc[kb][i] = (32 * kb + crow(i, hi) <= qrel) ? sacc[kb][i] : -INFINITY;
            }
            float mx0 = sacc[0][0], mx1 = sacc[1][0];
#pragma unroll
            for (int i = 1; i < 16; ++i) {
                mx0 = fmaxf(mx0, sacc[0][i]);
                mx1 = fmaxf(mx1, sacc[1][i]);
            }
            float mx = fmaxf(mx0, mx1);
            {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                mx = fmaxf(fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])), mx);  // own value and lane ^ 32's
            }
            const float mnew = fmaxf(m, mx);
            const float alpha = (ABL & 4) ? (m - mnew) : __builtin_amdgcn_exp2f((m - mnew) * sm_scale2);
            m = mnew;
            const float nmc = -mnew * sm_scale2;
            float rs4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float x = __builtin_fmaf(sacc[kb][i], sm_scale2, nmc);  // masked scores: 2^(-inf) = 0
                    const float e = (ABL & 4) ? x : __builtin_amdgcn_exp2f(x);
                    sacc[kb][i] = e;
                    rs4[i & 3] += e;
                }
            float rs = (rs4[0] + rs4[1]) + (rs4[2] + rs4[3]);
            {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(rs), __float_as_uint(rs), false, false);
                rs = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);  // one of the two is this lane's own sum, the other lane ^ 32's
            }
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
            // ---- O += P . V over four 16-key k-steps; P as an exact hi + lo pair (mask, subtract, v_cvt_pkrtz: two MFMAs per block) -----------
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                typedef __fp16 pk_h2 __attribute__((ext_vector_type(2)));
                h8 pa, pl;
                {
                    uint32_t hw[4], lw[4];
#pragma unroll
                    for (int q2 = 0; q2 < 4; ++q2) {
                        const float p0 = sacc[s >> 1][8 * (s & 1) + 2 * q2], p1 = sacc[s >> 1][8 * (s & 1) + 2 * q2 + 1];
                        const float h0 = __uint_as_float(__float_as_uint(p0) & 0xffffe000u), h1 = __uint_as_float(__float_as_uint(p1) & 0xffffe000u);
                        hw[q2] = __builtin_bit_cast(uint32_t, (pk_h2)__builtin_amdgcn_cvt_pkrtz(h0, h1));
                        lw[q2] = __builtin_bit_cast(uint32_t, (pk_h2)__builtin_amdgcn_cvt_pkrtz(p0 - h0, p1 - h1));
                    }
                    pa = __builtin_bit_cast(h8, make_uint4(hw[0], hw[1], hw[2], hw[3]));
                    pl = __builtin_bit_cast(h8, make_uint4(lw[0], lw[1], lw[2], lw[3]));
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint16_t* sub = Vs + (s * (D / 16) + 2 * c + chblk) * P3_VSUB;  // keys 16 s .. + 16, channels 32 c + 16 chblk .. + 16
                    const uint2 v0 = p3_v_frag(sub, 4 * hi, l15), v1 = p3_v_frag(sub, 8 + 4 * hi, l15);
                    const h8 bv = __builtin_bit_cast(h8, make_uint4(v0.x, v0.y, v1.x, v1.y));
                    if (!(ABL & 1)) o[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl, bv, o[c], 0, 0, 0);
                    o[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa, bv, o[c], 0, 0, 0);
                }
            }
        }
    }
    // ---- epilogue: O / l, fp16; value i of channel block c = query wrow0 + crow(i, hi), channel 32 c + l31 ----------------------------
    if (!wave_active) return;
    if constexpr (SPLIT) {
        const float to_nat = 1.0f / sqrtf((float)D);  // the running maximum is in raw score units; the merge works on natural-log scores
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = crow(i, hi);
            const float lr = __shfl(l, r, 64), mr = __shfl(m, r, 64);
            const int64_t qrow_i = wrow0 + r;
            if (qrow_i < seqlen) {
                float* wrow = ws + (((seq_starts[b] + qrow_i - ws_row0) * H + hq) * (int64_t)gridDim.y + blockIdx.y) * (D + 2);
#pragma unroll
                for (int c = 0; c < 4; ++c) wrow[c * 32 + l31] = o[c][i];
                if (l31 == 0) { wrow[D] = mr * to_nat; wrow[D + 1] = lr; }
            }
        }
        return;
    }
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

// merges the splits of token rows [row0, row0 + gridDim.x / H): one block per (row, head), one thread per channel
__global__ void attn_prefill32_reduce_kernel(const float* __restrict__ ws, int nsplit, int64_t row0, int H, uint16_t* __restrict__ out) {
    constexpr int D = P3_D;
    const int64_t bh = row0 * H + blockIdx.x;  // absolute (row, head) of the output; the workspace is relative to row0
    const float* w = ws + (int64_t)blockIdx.x * nsplit * (D + 2);
    const int d = threadIdx.x;
    float mm = -1e30f;
    for (int sp = 0; sp < nsplit; ++sp) mm = fmaxf(mm, w[sp * (D + 2) + D]);
    float ll = 0.f, o = 0.f;
    for (int sp = 0; sp < nsplit; ++sp) {
        const float a = __expf(w[sp * (D + 2) + D] - mm);
        ll = fmaf(w[sp * (D + 2) + D + 1], a, ll);
        o = fmaf(w[sp * (D + 2) + d], a, o);
    }
    out[bh * D + d] = f2h(o / ll);
}

#undef P3_LOAD_TILE
#undef P3_LOAD_ITEM
#undef P3_STORE_ITEM
#undef P3_ITEM_KEY
#undef P3_ITEM_PC

}  // namespace

hipError_t launch_attn_prefill32(hipStream_t s, const uint16_t* qkv, const KvAddr& kv, int quant_bit, const int64_t* seq_starts,
                                 const int64_t* start_pos, const int64_t* cache_indices, int64_t max_pages, int64_t b0, int64_t B,
                                 int H, int Hkv, int D, int64_t max_seq_len, uint16_t* out, int64_t max_kv_len, float* ws, size_t ws_bytes,
                                 int64_t row0, int64_t nrows) {
    if (D != P3_D || (quant_bit != 0 && quant_bit != 8)) return hipErrorInvalidValue;
    if (B <= b0 || max_seq_len <= 0) return hipSuccess;
    // Few new tokens behind long caches (a prefix-cache hit recomputes one page; a short follow-up turn): when the launch has too few
    // (query block, request, head) blocks to fill the chip, the keys of each are split over gridDim.y blocks whose partial rows are merged
    // (flash-decoding form).  row0 / nrows: the token rows of these requests (contiguous in the step); ws: nrows x H x splits x (D + 2) floats.
    static const int split_env = getenv("PPLHIP_P32_SPLIT") ? atoi(getenv("PPLHIP_P32_SPLIT")) : -1;  // 0: never; n > 1: force n
    const int64_t nqb4 = (max_seq_len + 127) / 128, blocks4 = nqb4 * (B - b0) * H;
    if (split_env != 0 && ws && nrows > 0 && max_seq_len < 1024 && blocks4 < 256 && max_kv_len >= 1024) {
        const int64_t ntiles = (max_kv_len + P3_BN - 1) / P3_BN;
        int64_t nsplit = (512 + blocks4 - 1) / blocks4;
        if (nsplit > ntiles / 4) nsplit = ntiles / 4;   // at least four 64-key tiles per split
        if (nsplit > 32) nsplit = 32;
        if (split_env > 1) nsplit = split_env;
        while (nsplit > 1 && (size_t)nrows * H * nsplit * (P3_D + 2) * sizeof(float) > ws_bytes) --nsplit;
        if (nsplit > 1) {
            const int nreq = (int)(B - b0);
            dim3 grid((unsigned)blocks4, (unsigned)nsplit);
#define P3_SPLIT(QB, MD) hipLaunchKernelGGL((attn_prefill32_kernel<QB, MD, 4, 0, 1>), grid, dim3(256), 0, s, qkv, kv, seq_starts, start_pos, \
                                            cache_indices, max_pages, b0, H, Hkv, nreq, (int)nqb4, out, ws, row0)
            if (quant_bit == 8) { if (kv.mode == 0) P3_SPLIT(8, 0); else P3_SPLIT(8, 1); }
            else { if (kv.mode == 0) P3_SPLIT(0, 0); else P3_SPLIT(0, 1); }
#undef P3_SPLIT
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(attn_prefill32_reduce_kernel, dim3((unsigned)(nrows * H)), dim3(P3_D), 0, s, ws, (int)nsplit, row0, H, out);
            return hipGetLastError();
        }
    }
    // 8 waves (256 query rows per block: each staged K / V tile serves twice the rows) once the sequences are long enough: 8192 new
    // tokens 1002 vs 1043 us, 2048 over a 6144-token cache 481 vs 538 us, but 16 x 512 tokens 105 vs 101 us
    static const int nw_env = tune_int("PPLHIP_P32_NW", 0);
    const int nw = nw_env ? nw_env : (max_seq_len >= 1024 ? 8 : 4);
    const int bm = nw * 32;
    const int nqb = (int)((max_seq_len + bm - 1) / bm), nreq = (int)(B - b0);
    dim3 grid((unsigned)((int64_t)nqb * nreq * H));
#define P3_LAUNCH(QB, MD, NW_) hipLaunchKernelGGL((attn_prefill32_kernel<QB, MD, NW_>), grid, dim3(NW_ * 64), 0, s, qkv, kv, seq_starts, start_pos, \
                                                  cache_indices, max_pages, b0, H, Hkv, nreq, nqb, out, nullptr, 0)
#ifdef P3_ABLATE_BUILD  // diagnosis build (profiles/r03_prefill_attention_ablation.md): wrong results, same instruction stream otherwise
    static const int abl = getenv("PPLHIP_P32_ABLATE") ? atoi(getenv("PPLHIP_P32_ABLATE")) : 0;   // (an ablation build reads its switch itself: no TUNING=1 needed)
#define P3_ABL(A) if (abl == A && nw == 8) { hipLaunchKernelGGL((attn_prefill32_kernel<8, 0, 8, A>), grid, dim3(512), 0, s, qkv, kv, seq_starts, start_pos, cache_indices, max_pages, b0, H, Hkv, nreq, nqb, out, nullptr, 0); return hipGetLastError(); }
    P3_ABL(1) P3_ABL(2) P3_ABL(3) P3_ABL(7)
#undef P3_ABL
#endif
    if (nw == 8) {
        if (quant_bit == 8) { if (kv.mode == 0) P3_LAUNCH(8, 0, 8); else P3_LAUNCH(8, 1, 8); }
        else { if (kv.mode == 0) P3_LAUNCH(0, 0, 8); else P3_LAUNCH(0, 1, 8); }
    } else {
        if (quant_bit == 8) { if (kv.mode == 0) P3_LAUNCH(8, 0, 4); else P3_LAUNCH(8, 1, 4); }
        else { if (kv.mode == 0) P3_LAUNCH(0, 0, 4); else P3_LAUNCH(0, 1, 4); }
    }
#undef P3_LAUNCH
    return hipGetLastError();
}

}  // namespace pplhip
