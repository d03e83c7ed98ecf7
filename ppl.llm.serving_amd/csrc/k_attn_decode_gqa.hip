// K8, grouped-query variant (4 <= H / Hkv <= 16, e.g. LLaMA-2-70B: 8 query heads per KV head).  The VALU decode kernel
// (k_attn_decode.hip) would read every KV row once per query head -- 8x the bytes -- so decode rows of grouped-query models
// run on the matrix cores with the head group as the MFMA's second dimension: every KV byte is read from HBM ONCE.
//
// Round 3 rewrite.  HBM-bound kernel, so the matrix cores and the VALU have time to spare -- spent on two things:
//   * precision: the round-2 kernel rounded the dequantised K / V (int8 x fp16 scale: 19 significant bits) and the
//     probabilities to fp16 for the MFMAs and landed 4e-3 from the oracle at kv 2048 (70B / TP8 geometry), six times the
//     oracle's own summation-order noise.  Now every MFMA operand is an exact hi + lo pair of fp16 numbers:
//       K (V) : hi = fp16(q * s) (packed multiply), lo = fma(q, s, -hi) -- the error term of a floating-point product is itself
//              exactly representable (TwoProduct), so hi + lo == q * s exactly, for 1.5 packed VALU ops per element
//              (V: with GQ_V_EXACT = 1, the default; see below);
//       P    : hi = fp16(p), lo = fp16(p - hi) (22 significant bits);
//     S = Q.(Khi + Klo)  (two MFMAs per k-step), O += (Phi + Plo).(Vhi + Vlo).  The P.V MFMA contracts over 32 k-slots but
//     a wave's sub-tile has only 16 keys, so the lo terms ride in the k-slots that used to be zero: A = (Phi | Plo),
//     B = (Vhi | Vhi) gives Phi.Vhi + Plo.Vhi in ONE instruction; int8 KV adds A = (Phi | Plo), B = (Vlo | Vlo).
//     fp32 accumulation throughout: what is left against the oracle is summation order.
//   * no workgroup barriers in the streaming loop: a wave owns whole 16-key sub-tiles (keys tbeg + 16 (8 t + wave)), loads K
//     STRAIGHT from HBM into the MFMA A-operand layout (lane = (key, quarter) takes 16-byte pieces quarter, quarter + 4, ...; the
//     matching permutation of the contraction index is applied to the Q fragments), and passes V through a wave-PRIVATE LDS
//     region only to transpose it (ds_read_b64_tr_b16).  GQ_NBUF sub-tiles are in flight per wave in a statically rotating set of register
//     buffers; the 8 waves of a block meet once, to merge their online-softmax states.
// Round 6: the two register buffers are consumed as ONE 32-key step (pair_step below) -- a SIMD issues about one instruction per 8.5 cycles
// from this dependent VALU / LDS / MFMA mix however many waves it holds, so the kernel's time is its instruction count
// (profiles/r06_gqa_experiments.md): -23 % VALU per key, +13 % bandwidth at config 4's shape.
// Grid (Hkv, requests, splits); workspace / reduce kernel shared with the multi-head kernel (k_attn_decode.hip).
// Oracle: ref_attention (oracle/llama_ref.c).
#include <stdlib.h>
#include <mutex>
#include <set>
#include <type_traits>
#include <utility>
#include <hip/hip_ext.h>
#include "kernels.h"

namespace pplhip {

namespace {

#ifndef GQ_THREADS_N
#define GQ_THREADS_N 512
#endif
// waves per block: GQ_THREADS_N / 64 when a launch has fewer than GQ_SMALL_BLOCK_MIN blocks (config 4: 256 requests x 1 KV head = one 8-wave block per
// CU), else 4 -- two blocks per CU (70 KiB of LDS each), so that one block's ramp (page ids, q, first K / V loads) and its merge run beside the
// other's stream: short contexts in big batches are mostly ramp and merge (B 1024 x kv 512: 51 us for 21 us of traffic with 8-wave blocks)
constexpr int GQ_WAVES_BIG = GQ_THREADS_N / 64, GQ_WAVES_SMALL = 4;
#ifndef GQ_SMALL_BLOCK_MIN
#define GQ_SMALL_BLOCK_MIN 512
#endif
#ifndef GQ_NBUF
#define GQ_NBUF 2   // 16-key sub-tiles in flight per wave (register buffers): one PAIR (2) or two (4); round 3 measured 2 / 3 / 4 single steps at 4.46 / 4.07 / 4.16 TB/s, round 6 two pairs against one: equal at kv 2048, 10-15 % slower below (profiles/r06_gqa_nbuf_ab.log) -- issue-bound, not latency-bound
#endif
// GQ_V_EXACT 1 (default again in round 5): V as an exact hi + lo pair like K and P.  Round 4 took V as int8 x scale rounded to fp16 ONCE
// (one LDS image, one MFMA per channel block, 16 packed VALU ops less per sub-tile: config 4 4.48-4.62 -> 4.84 TB/s) for 6.4e-5 -> 2.6e-4
// of max|out| at the operator -- and that term turned out to be what put the 70B / TP8 W4A16 model case at 1.35e-3 of max|logit| = 1.9 x
// the oracle's own noise floor and 0.90 of its cap (profiles/r05_w4_gqa_margin.log: 0.57e-3 = 0.8 x the floor with V exact, same
// library otherwise), which in turn kept the faster RMSNorm form switched off.  Precision first: exact V, and the RMSNorm form pays the
// time back (config 4 per rank: +0.25 ms and -0.25 ms).  0: the rounded form (build switch).
#ifndef GQ_V_EXACT
#define GQ_V_EXACT 1
#endif
constexpr int GQ_VSUB = 272;  // halfs per [16 keys][16 channels] V sub-tile in LDS: 256 + 16 of skew (bank spread of the writes)

typedef short gq_s4 __attribute__((__vector_size__(4 * sizeof(short))));
// transposing LDS read of a row-major [16 keys][16 channels] fp16 sub-tile: lane (channel l15, quarter kq) receives keys
// kq*4 .. kq*4+3 of channel l15 (lane semantics pinned by profiles/probes/lds_tr_read_probe.hip)
__device__ __forceinline__ uint2 gq_v_frag(const uint16_t* sub, int kq, int l15) {
    const uint16_t* p = sub + (kq * 4 + (l15 >> 2)) * 16 + (l15 & 3) * 4;
    const gq_s4 w = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) gq_s4*)p);
    return __builtin_bit_cast(uint2, w);
}

__device__ __forceinline__ h8 splat8(_Float16 v) { return h8{v, v, v, v, v, v, v, v}; }

// exact product q * s of 8 small integers (fp16-exact) and an fp16 scale as hi + lo
__device__ __forceinline__ void two_product(h8 q, _Float16 s, h8& hi, h8& lo) {
    const h8 sv = splat8(s);
    hi = q * sv;
    lo = __builtin_elementwise_fma(q, sv, -hi);
}

template <int QBIT, int D, int GQ_WAVES>
constexpr int gq_lds_bytes() {
    constexpr int nimg = (QBIT == 8 && GQ_V_EXACT) ? 2 : 1;
    constexpr int v_bytes = GQ_WAVES * 2 * nimg * (D / 16) * GQ_VSUB * 2, merge_bytes = GQ_WAVES * 16 * (D + 2) * 4;
    return v_bytes > merge_bytes ? v_bytes : merge_bytes;
}

template <int QBIT, int D, int MODE, int GQ_WAVES>
__global__ __launch_bounds__(GQ_WAVES * 64) void attn_decode_gqa_kernel(const uint16_t* __restrict__ qkv, KvAddr kv,
                                                                     const int64_t* __restrict__ seq_starts,
                                                                     const int64_t* __restrict__ start_pos,
                                                                     const int64_t* __restrict__ cache_indices,
                                                                     int64_t max_pages, int H, int Hkv, int split,
                                                                     float* __restrict__ workspace, uint16_t* __restrict__ out) {
    constexpr int ELT = QBIT == 8 ? 1 : 2;
    constexpr int CH = 16 / ELT;        // channels in one 16-byte piece
    constexpr int LPT = D / CH;         // pieces per row
    constexpr int PPL = LPT / 4;        // pieces per lane and 16-key sub-tile (K and V alike)
    constexpr int KSTEPS = D / 32;
    constexpr int DT = D / 16;
    constexpr int NIMG = (QBIT == 8 && GQ_V_EXACT) ? 2 : 1;           // V images in LDS: hi (+ lo)
    constexpr int VW = DT * GQ_VSUB;                  // halfs of one image of one wave's 16-key sub-tile
    static_assert(LPT % 4 == 0, "a row must hold a multiple of four 16-byte pieces");
    static_assert(GQ_NBUF == 2 || GQ_NBUF == 4, "a wave consumes its sub-tiles in pairs: register buffers (0, 1)[, (2, 3)]");
    extern __shared__ __attribute__((aligned(16))) char smem[];   // gq_lds_bytes<QBIT, D>(): per wave two sub-tiles' V image(s); the merge buffer overlays them

    const int hk = blockIdx.x;
    const int64_t b = blockIdx.y;
    const int sp_i = blockIdx.z;
    const int grp = H / Hkv;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave index in an SGPR
    const int l15 = lane & 15, kq = lane >> 4;
    uint16_t* const vw = reinterpret_cast<uint16_t*>(smem) + wave * 2 * NIMG * VW;  // this wave's private V image(s): sub-tile 0, then sub-tile 1
    const int64_t rowstride = (int64_t)(H + 2 * Hkv) * D;
    const int64_t kv_len = start_pos[b] + 1;
    const int64_t per = ((kv_len + split - 1) / split + 127) / 128 * 128;  // whole 128-key strips per split
    const int64_t tbeg = sp_i * per;
    const int64_t tend = (tbeg + per < kv_len) ? tbeg + per : kv_len;

    // Q fragments (MFMA B operand): lane (n = head l15 of the group, kq), k-slots of step ks = the 8 channels the K fragment of
    // the same (ks, kq) holds: int8 piece p = kq + 4 (ks / 2) -> channels 16 p + 8 (ks % 2); fp16 piece p = kq + 4 ks -> 8 p
    h8 qf[KSTEPS];
    {
        const uint16_t* qrow = qkv + seq_starts[b] * rowstride + (int64_t)(hk * grp + (l15 < grp ? l15 : 0)) * D;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const int ch = QBIT == 8 ? 16 * (kq + 4 * (ks >> 1)) + 8 * (ks & 1) : 8 * (kq + 4 * ks);
            const uint4 v = *reinterpret_cast<const uint4*>(qrow + ch);
            qf[ks] = __builtin_bit_cast(h8, l15 < grp ? v : make_uint4(0, 0, 0, 0));
        }
    }
    f4 o[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i) o[i] = f4{0.f, 0.f, 0.f, 0.f};
    float m = -1e30f, l = 0.f;
    const float sm_scale = 1.0f / sqrtf((float)D);

    const int64_t slot0 = MODE == 0 ? cache_indices[b] : 0;
    const char* kbase = reinterpret_cast<const char*>(kv.cache) + (int64_t)hk * kv.sH * ELT;
    const char* vbase = kbase + kv.sKV * ELT;
    const uint16_t* ksbase = kv.scale + (int64_t)hk * kv.ssH;
    const uint16_t* vsbase = ksbase + kv.ssKV;

    // V staging item j of this lane: (key, piece) = divmod(lane + 64 j, LPT): a load instruction covers whole rows
    constexpr int NBUF = GQ_NBUF;
    uint4 kraw[NBUF][PPL], vraw[NBUF][PPL];
    uint32_t ksc[NBUF][PPL], vsc[NBUF][PPL];
    // Addressing: a sub-tile is 16 consecutive keys starting at a multiple of 16, wave-uniform; with contiguous slots, or pages
    // whose size is a multiple of 16, its rows are consecutive slots, so the row base is ONE scalar computation per sub-tile and a
    // lane adds its constant (key-in-sub-tile x row pitch + piece) -- instead of a 64-bit multiply chain per load
    const int64_t rowb = kv.sN * ELT, srow = kv.ssN;                 // row pitch of the cache (bytes) and of the scales (halfs)
    const int rowb32 = (int)rowb, srow32 = (int)srow;                // a sub-tile spans 16 rows: the lane part fits 32 bits
    const bool uniform_rows = MODE == 0 || (kv.page_size % 16 == 0);
    int vkey_l[PPL], vpc_l[PPL];
#pragma unroll
    for (int j = 0; j < PPL; ++j) { vkey_l[j] = (lane + 64 * j) / LPT; vpc_l[j] = (lane + 64 * j) % LPT; }
    // paged cache: the page id of the wave's NEXT sub-tile is fetched (one scalar load) while the current one is being loaded, so the
    // page-table latency never sits in front of the KV loads (the wave's sub-tiles are GQ_WAVES x 16 keys apart, in call order)
    const int64_t* const page_row = cache_indices + b * max_pages;
    auto page_of = [&](int64_t kb) -> int64_t {
        const int64_t kbc = kb < tend ? kb : ((tend - 1) & ~(int64_t)15);
        return kv.page_shift >= 0 ? (kbc >> kv.page_shift) : kbc / kv.page_size;
    };
    int64_t pid_next = 0;
    if (MODE == 1 && uniform_rows && tbeg < tend) pid_next = page_row[page_of(tbeg + (int64_t)wave * 16)];
    auto load_sub = [&](int p, int64_t kb) {
        // prefetches past the range re-read the range's last sub-tile (never consumed); rows past `tend` inside the last
        // sub-tile are clamped to its last valid row and masked in the softmax
        const int64_t kbc = kb < tend ? kb : ((tend - 1) & ~(int64_t)15);
        const int last = (int)(tend - 1 - kbc);                      // >= 0
        const int kk = l15 < last ? l15 : last;
        if (uniform_rows) {
            int64_t slot_b;
            if (MODE == 0) slot_b = slot0 + kbc;
            else {
                const int64_t pg = kv.page_shift >= 0 ? (kbc >> kv.page_shift) : kbc / kv.page_size;
                slot_b = pid_next * kv.page_size + (kbc - pg * kv.page_size);
                pid_next = page_row[page_of(kb + 16 * GQ_WAVES)];
            }
            const char* kp = kbase + slot_b * rowb;
            const char* vp = vbase + slot_b * rowb;
            const uint16_t* ksp = ksbase + slot_b * srow;
            const uint16_t* vsp = vsbase + slot_b * srow;
#pragma unroll
            // (unsigned 32-bit lane offsets on a wave-uniform base: the loads take the scalar-base form instead of a 64-bit VALU add each)
            for (int j = 0; j < PPL; ++j) {
                const int pc = kq + 4 * j;
                kraw[p][j] = kv_stream_load(reinterpret_cast<const uint4*>(kp + (uint32_t)(kk * rowb32 + pc * 16)));
                if constexpr (QBIT == 8) ksc[p][j] = kv_stream_load(reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(ksp) + (uint32_t)((kk * srow32 + pc * 2) * 2)));
            }
#pragma unroll
            for (int j = 0; j < PPL; ++j) {
                const int vk = vkey_l[j] < last ? vkey_l[j] : last;
                vraw[p][j] = kv_stream_load(reinterpret_cast<const uint4*>(vp + (uint32_t)(vk * rowb32 + vpc_l[j] * 16)));
                if constexpr (QBIT == 8) vsc[p][j] = kv_stream_load(reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(vsp) + (uint32_t)((vk * srow32 + vpc_l[j] * 2) * 2)));
            }
        } else {  // pages smaller than (or not aligned to) a sub-tile: every row through the page table
            const int64_t kslot = kv_slot(kv, cache_indices, max_pages, b, kbc + kk);
#pragma unroll
            for (int j = 0; j < PPL; ++j) {
                const int pc = kq + 4 * j;
                kraw[p][j] = kv_stream_load(reinterpret_cast<const uint4*>(kbase + kslot * rowb + pc * 16));
                if constexpr (QBIT == 8) ksc[p][j] = kv_stream_load(reinterpret_cast<const uint32_t*>(ksbase + kslot * srow + pc * 2));
            }
#pragma unroll
            for (int j = 0; j < PPL; ++j) {
                const int vk = vkey_l[j] < last ? vkey_l[j] : last;
                const int64_t vslot = kv_slot(kv, cache_indices, max_pages, b, kbc + vk);
                vraw[p][j] = kv_stream_load(reinterpret_cast<const uint4*>(vbase + vslot * rowb + vpc_l[j] * 16));
                if constexpr (QBIT == 8) vsc[p][j] = kv_stream_load(reinterpret_cast<const uint32_t*>(vsbase + vslot * srow + vpc_l[j] * 2));
            }
        }
    };

    // Round 6: a wave consumes its sub-tiles in PAIRS -- register buffers 0 and 1, keys kb .. + 16 and kb + 16 GQ_WAVES .. + 16 -- as ONE 32-key step:
    // both score tiles first (two independent MFMA chains), ONE online-softmax update over the 32 keys (one pair of cross-lane maxima and
    // sums, one rescale of the output accumulators instead of two), and P.V 32 deep: the MFMA's k-slots take four keys of each sub-tile,
    // A = (P_hi of sub-tile 0 | P_hi of sub-tile 1) and then the same for P_lo, B = (V of sub-tile 0 | V of sub-tile 1) straight out of two
    // transposing reads.  Until round 5 a 16-key step put (P_hi | P_lo) against (V | V): the duplicated V operand cost two v_mov per
    // fragment (36 of ~330 VALU instructions per sub-tile), the softmax bookkeeping ran twice as often, and every sub-tile was one serial
    // chain (profiles/r06_gqa_experiments.md).  Same products, fp32 accumulation, lo terms first; a sub-tile past the range is all mask.
    uint16_t* const vw1 = vw + NIMG * VW;   // the second sub-tile's V image(s)
    auto stage_v = [&](auto ptag, uint16_t* dstw) {
        constexpr int P = decltype(ptag)::value;
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            const int key = vkey_l[j], ch0 = vpc_l[j] * CH;
            if constexpr (QBIT == 8) {
                const h8 q0 = cvt_i8x8_f16(make_uint2(vraw[P][j].x, vraw[P][j].y)), q1 = cvt_i8x8_f16(make_uint2(vraw[P][j].z, vraw[P][j].w));
                const h2 sc = __builtin_bit_cast(h2, vsc[P][j]);
                uint16_t* dst = dstw + (ch0 >> 4) * GQ_VSUB + key * 16;  // CH = 16: the piece is one whole sub-tile row
                if constexpr (GQ_V_EXACT) {
                    h8 hi0, lo0, hi1, lo1;
                    two_product(q0, sc[0], hi0, lo0);
                    two_product(q1, sc[1], hi1, lo1);
                    *reinterpret_cast<uint4*>(dst) = __builtin_bit_cast(uint4, hi0);
                    *reinterpret_cast<uint4*>(dst + 8) = __builtin_bit_cast(uint4, hi1);
                    *reinterpret_cast<uint4*>(dst + VW) = __builtin_bit_cast(uint4, lo0);
                    *reinterpret_cast<uint4*>(dst + VW + 8) = __builtin_bit_cast(uint4, lo1);
                } else {
                    *reinterpret_cast<uint4*>(dst) = __builtin_bit_cast(uint4, q0 * splat8(sc[0]));
                    *reinterpret_cast<uint4*>(dst + 8) = __builtin_bit_cast(uint4, q1 * splat8(sc[1]));
                }
            } else {
                *reinterpret_cast<uint4*>(dstw + (ch0 >> 4) * GQ_VSUB + key * 16 + (ch0 & 15)) = vraw[P][j];
            }
        }
    };
    // S^T = K . Q^T of one sub-tile, K fragments straight from the raw registers
    auto scores = [&](auto ptag) -> f4 {
        constexpr int P = decltype(ptag)::value;
        f4 sacc = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            if constexpr (QBIT == 8) {
                const h8 q0 = cvt_i8x8_f16(make_uint2(kraw[P][j].x, kraw[P][j].y)), q1 = cvt_i8x8_f16(make_uint2(kraw[P][j].z, kraw[P][j].w));
                const h2 sc = __builtin_bit_cast(h2, ksc[P][j]);
                h8 hi0, lo0, hi1, lo1;
                two_product(q0, sc[0], hi0, lo0);
                two_product(q1, sc[1], hi1, lo1);
                sacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(lo0, qf[2 * j], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(lo1, qf[2 * j + 1], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi0, qf[2 * j], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi1, qf[2 * j + 1], sacc, 0, 0, 0);
            } else {
                sacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, kraw[P][j]), qf[j], sacc, 0, 0, 0);
            }
        }
        return sacc;
    };
    constexpr int64_t STRIDE = 16 * GQ_WAVES;
    auto pair_step = [&](auto t0, auto t1, int64_t kb) {
        using T0 = decltype(t0);
        using T1 = decltype(t1);
        stage_v(T0{}, vw);
        stage_v(T1{}, vw1);
        f4 s0 = scores(T0{}), s1 = scores(T1{});
        load_sub(T0::value, kb + NBUF * STRIDE);   // the raw registers are free: the pair NBUF / 2 steps ahead
        load_sub(T1::value, kb + (NBUF + 1) * STRIDE);
        // ---- online softmax over the pair; this lane: head l15, keys kb + kq*4 + r (sub-tile 0) and kb + STRIDE + kq*4 + r (sub-tile 1)
        float mx = -1e30f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t k0 = kb + kq * 4 + r, k1 = k0 + STRIDE;
            s0[r] = (k0 < tend) ? s0[r] * sm_scale : -1e30f;
            s1[r] = (k1 < tend) ? s1[r] * sm_scale : -1e30f;
            mx = fmaxf(mx, fmaxf(s0[r], s1[r]));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(m, mx);
        const float alpha = __expf(m - mnew);
        m = mnew;
        float rs = 0.f;
        h8 p_hi, p_lo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t k0 = kb + kq * 4 + r, k1 = k0 + STRIDE;
            const float e0 = (k0 < tend) ? __expf(s0[r] - mnew) : 0.f;
            const float e1 = (k1 < tend) ? __expf(s1[r] - mnew) : 0.f;
            rs += e0 + e1;
            const _Float16 h0 = to_h(e0), h1 = to_h(e1);
            p_hi[r] = h0; p_hi[4 + r] = h1;
            p_lo[r] = to_h(e0 - (float)h0);       // the part of p that fp16 dropped
            p_lo[4 + r] = to_h(e1 - (float)h1);
        }
        rs += __shfl_xor(rs, 16, 64);
        rs += __shfl_xor(rs, 32, 64);
        l = l * alpha + rs;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
            float ar[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) ar[r] = __shfl(alpha, kq * 4 + r, 64);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[dt][r] *= ar[r];
        }
        // ---- O += P . V: V^T fragments through the transposing LDS read (same wave wrote them: program order, no barrier) ------
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const uint2 a = gq_v_frag(vw + dt * GQ_VSUB, kq, l15), b = gq_v_frag(vw1 + dt * GQ_VSUB, kq, l15);  // keys kq*4 .. +4 of channel dt*16 + l15, both sub-tiles
            if constexpr (QBIT == 8 && GQ_V_EXACT) {
                const uint2 al = gq_v_frag(vw + VW + dt * GQ_VSUB, kq, l15), bl = gq_v_frag(vw1 + VW + dt * GQ_VSUB, kq, l15);
                const h8 vl = __builtin_bit_cast(h8, make_uint4(al.x, al.y, bl.x, bl.y));
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(p_lo, vl, o[dt], 0, 0, 0);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(p_hi, vl, o[dt], 0, 0, 0);
            }
            const h8 vh = __builtin_bit_cast(h8, make_uint4(a.x, a.y, b.x, b.y));
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(p_lo, vh, o[dt], 0, 0, 0);
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(p_hi, vh, o[dt], 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the reads above precede the next pair's writes
        __builtin_amdgcn_wave_barrier();
    };

    const int64_t kb0 = tbeg + wave * 16;
    // the prefetches are unconditional (keys past the range are clamped and re-read the last row), so the number of loads in
    // flight is static at every wait
    if (kb0 < tend) {
#pragma unroll
        for (int p = 0; p < NBUF; ++p) load_sub(p, kb0 + p * STRIDE);
        for (int64_t kb = kb0; kb < tend; kb += NBUF * STRIDE) {
            pair_step(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, kb);
            if constexpr (NBUF == 4) {
                if (kb + 2 * STRIDE >= tend) break;
                pair_step(std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{}, kb + 2 * STRIDE);
            }
        }
    }
    __syncthreads();  // every wave is done with its V region: the merge buffer may overlay them
    // ---- merge the 8 waves: partial (o[head][d], m[head], l[head]) per wave through LDS ------------------------
    float* mg = reinterpret_cast<float*>(smem);  // [wave][16 heads][D + 2]
    {
        float mr[4], lr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            mr[r] = __shfl(m, kq * 4 + r, 64);
            lr[r] = __shfl(l, kq * 4 + r, 64);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float* row = mg + ((wave * 16) + kq * 4 + r) * (D + 2);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) row[dt * 16 + l15] = o[dt][r];
            if (l15 == 0) { row[D] = mr[r]; row[D + 1] = lr[r]; }
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < grp * D; idx += GQ_WAVES * 64) {
        const int head = idx / D, d = idx - head * D;
        float mm = -1e30f;
        for (int w = 0; w < GQ_WAVES; ++w) mm = fmaxf(mm, mg[(w * 16 + head) * (D + 2) + D]);
        float ll = 0.f, ov = 0.f;
        for (int w = 0; w < GQ_WAVES; ++w) {
            const float* row = mg + (w * 16 + head) * (D + 2);
            const float a = __expf(row[D] - mm);
            ll = fmaf(row[D + 1], a, ll);
            ov = fmaf(row[d], a, ov);
        }
        const int hq = hk * grp + head;
        if (split == 1) {
            out[(b * H + hq) * (int64_t)D + d] = f2h(ov / ll);
        } else {
            float* ws = workspace + ((b * H + hq) * (int64_t)split + sp_i) * (D + 2);
            ws[d] = ov;
            if (d == 0) { ws[D] = mm; ws[D + 1] = ll; }
        }
    }
}

}  // namespace

// more than 64 KiB of LDS per block (two sub-tiles' exact V images per wave): the attribute belongs to the function on the current device
static void gq_set_lds(const void* fn, int bytes) {
    if (bytes <= 65536) return;
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(mu);
    if (done.insert({fn, dev}).second) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

bool attn_decode_gqa_supported(int quant_bit, int H, int Hkv, int D) {
    if (Hkv <= 0 || H % Hkv) return false;
    const int grp = H / Hkv;
    if (grp < 4 || grp > 16) return false;
    if (quant_bit == 8) return D == 128 || D == 64;  // a row must hold >= four 16-byte pieces
    return quant_bit == 0 && (D == 128 || D == 64 || D == 32);
}

hipError_t launch_attn_decode_gqa(hipStream_t s, const uint16_t* qkv, const KvAddr& kv, int quant_bit,
                                  const int64_t* seq_starts, const int64_t* start_pos, const int64_t* cache_indices,
                                  int64_t max_pages, int64_t nb, int H, int Hkv, int D, int split, float* workspace,
                                  uint16_t* out, hipEvent_t t0, hipEvent_t t1) {
    if (nb == 0) return hipSuccess;
    if (!attn_decode_gqa_supported(quant_bit, H, Hkv, D)) return hipErrorInvalidValue;
    dim3 grid((unsigned)Hkv, (unsigned)nb, (unsigned)split);
    static const int small_min = getenv("PPLHIP_GQA_SMALL_BLOCK_MIN") ? atoi(getenv("PPLHIP_GQA_SMALL_BLOCK_MIN")) : GQ_SMALL_BLOCK_MIN;   // A/B runs
    const bool small = (int64_t)Hkv * nb * split >= small_min;
#define GQ_LAUNCH(QB, DD, MD, NW)                                                                                                 \
    do {                                                                                                                          \
        gq_set_lds((const void*)attn_decode_gqa_kernel<QB, DD, MD, NW>, gq_lds_bytes<QB, DD, NW>());                             \
        if (t0 && t1)                                                                                                             \
            hipExtLaunchKernelGGL((attn_decode_gqa_kernel<QB, DD, MD, NW>), grid, dim3(NW * 64), (gq_lds_bytes<QB, DD, NW>()), s, t0, t1, 0, qkv, kv, \
                                  seq_starts, start_pos, cache_indices, max_pages, H, Hkv, split, workspace, out);               \
        else                                                                                                                      \
            hipLaunchKernelGGL((attn_decode_gqa_kernel<QB, DD, MD, NW>), grid, dim3(NW * 64), (gq_lds_bytes<QB, DD, NW>()), s, qkv, kv, seq_starts, \
                               start_pos, cache_indices, max_pages, H, Hkv, split, workspace, out);                              \
    } while (0)
#define GQ_CASE(QB, DD)                                                                                          \
    if (quant_bit == QB && D == DD) {                                                                            \
        if (small) { if (kv.mode == 0) GQ_LAUNCH(QB, DD, 0, GQ_WAVES_SMALL); else GQ_LAUNCH(QB, DD, 1, GQ_WAVES_SMALL); } \
        else { if (kv.mode == 0) GQ_LAUNCH(QB, DD, 0, GQ_WAVES_BIG); else GQ_LAUNCH(QB, DD, 1, GQ_WAVES_BIG); }   \
        return hipGetLastError();                                                                                \
    }
    GQ_CASE(8, 128) GQ_CASE(0, 128) GQ_CASE(8, 64) GQ_CASE(0, 64) GQ_CASE(0, 32)
#undef GQ_CASE
#undef GQ_LAUNCH
    return hipErrorInvalidValue;
}

}  // namespace pplhip
