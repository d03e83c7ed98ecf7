// K4 + K5 fused: rotary position embedding on q and k of the fused qkv rows, then write of k, v into the KV
// slab -- with int8 group-8 quantisation when cache_quant_bit == 8.  HBM-bound; 1-4 workgroups per token.
//
// Work item = one (head, 8-channel block) of a token:
//   q / k heads: channels [i0, i0+8) and their RoPE partners [i0+D/2, i0+D/2+8) (half-split pairing) -- two
//                complete quantisation groups, so rotate + quantise + store needs no cross-lane traffic;
//   v heads    : channels [i0, i0+8), copy / quantise only.
// Position of row t of request b: start_pos[b] + (t - seq_starts[b]) (src/generator/llm_generator.cc:263-298);
// slot of (b, pos): kv_slot() (k_common.h).  Oracle: ref_rope_kv_write (oracle/llama_ref.c).
#include <stdlib.h>
#include "kernels.h"

namespace pplhip {

template <int QBIT>
__device__ __forceinline__ void store_group8(const KvAddr& kv, int kvsel, int head, int64_t slot, int ch0,
                                             const float* x /*8*/) {
    const int64_t base = (int64_t)kvsel * kv.sKV + (int64_t)head * kv.sH + slot * kv.sN + ch0;
    if constexpr (QBIT == 0) {
        *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(kv.cache) + base) = pack8(x);
    } else {
        float mx = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) mx = fmaxf(mx, fabsf(x[i]));
        const _Float16 sh = to_h(mx / 127.0f);
        const float sf = (float)sh;
        const float inv = sf > 0.f ? __fdiv_rn(1.0f, sf) : 0.f;  // one correctly rounded reciprocal per group
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float q = rintf(__fmul_rn(x[i], inv));
            q = fminf(fmaxf(q, -127.f), 127.f);
            const uint32_t b = (uint32_t)(int)q & 0xffu;
            if (i < 4) lo |= b << (8 * i); else hi |= b << (8 * (i - 4));
        }
        *reinterpret_cast<uint2*>(reinterpret_cast<int8_t*>(kv.cache) + base) = make_uint2(lo, hi);
        const int64_t sbase = (int64_t)kvsel * kv.ssKV + (int64_t)head * kv.ssH + slot * kv.ssN + ch0 / 8;
        kv.scale[sbase] = __builtin_bit_cast(uint16_t, sh);
    }
}

template <int QBIT>
__global__ __launch_bounds__(256) void rope_kv_write_kernel(uint16_t* __restrict__ qkv, const float* __restrict__ cos_sin,
                                                            KvAddr kv, const int64_t* __restrict__ seq_starts,
                                                            const int64_t* __restrict__ start_pos,
                                                            const int64_t* __restrict__ cache_indices, int64_t max_pages,
                                                            int64_t B, int64_t t0, int H, int Hkv, int D, SplitSlabs sl) {
    const int64_t t = t0 + blockIdx.x;
    // request of row t: last b with seq_starts[b] <= t.  Every thread finds it for itself (wave-uniform loads that hit L2): no LDS
    // broadcast and no barrier in front of the row's own loads, which depend on t alone and are issued beside this chain.  Decode rows
    // come first and have one token each, so row t usually IS request t: two independent loads instead of log2(B) dependent ones
    int64_t b;
    {
        const int64_t g = t < B ? t : B - 1;
        int64_t lo = 0, hi = B - 1;
        if (seq_starts[g] <= t && t < seq_starts[g + 1]) {
            lo = g;
        } else {
            while (lo < hi) {
                const int64_t mid = (lo + hi + 1) >> 1;
                if (seq_starts[mid] <= t) lo = mid; else hi = mid - 1;
            }
        }
        b = lo;
    }
    const int64_t pos = start_pos[b] + (t - seq_starts[b]);
    const int64_t slot = kv_slot(kv, cache_indices, max_pages, b, pos);
    const int half = D / 2;
    const int bph = half / 8;                  // rope work items per head
    const int n_rope = (H + Hkv) * bph;
    const int n_v = Hkv * (D / 8);
    uint16_t* row = qkv + t * (int64_t)(H + 2 * Hkv) * D;
    const float* cs = cos_sin + pos * D;       // cos[0..half) then sin[0..half)
    // gridDim.y blocks share a token's work items (one item per thread at the LLaMA geometries: the step is a chain of dependent
    // latencies, not bandwidth -- 7B, 128 rows: 11.8 -> see DESIGN.md)
    for (int w = blockIdx.y * 256 + threadIdx.x; w < n_rope + n_v; w += 256 * gridDim.y) {
        if (w < n_rope) {
            const int head = w / bph, i0 = (w - head * bph) * 8;
            uint16_t* x = row + (int64_t)head * D;
            float a[8], bb[8], c[8], s[8], ra[8], rb[8];
            if (sl.splits) {  // the qkv rows straight from wqkv's unreduced split-K slabs (kernels.h SplitSlabs)
                slab_load8(sl, t - t0, head * D + i0, a);   // (slab rows count from the launch's first row)
                slab_load8(sl, t - t0, head * D + i0 + half, bb);
            } else {
                unpack8(*reinterpret_cast<const uint4*>(x + i0), a);
                unpack8(*reinterpret_cast<const uint4*>(x + i0 + half), bb);
            }
            *reinterpret_cast<float4*>(c) = *reinterpret_cast<const float4*>(cs + i0);
            *reinterpret_cast<float4*>(c + 4) = *reinterpret_cast<const float4*>(cs + i0 + 4);
            *reinterpret_cast<float4*>(s) = *reinterpret_cast<const float4*>(cs + half + i0);
            *reinterpret_cast<float4*>(s + 4) = *reinterpret_cast<const float4*>(cs + half + i0 + 4);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                // two roundings per product pair as in the oracle: fp32 mul/sub, one fp16 rounding at the end
                ra[i] = round_h(__fsub_rn(__fmul_rn(a[i], c[i]), __fmul_rn(bb[i], s[i])));
                rb[i] = round_h(__fadd_rn(__fmul_rn(bb[i], c[i]), __fmul_rn(a[i], s[i])));
            }
            if (head < H) {  // q: in place
                *reinterpret_cast<uint4*>(x + i0) = pack8(ra);
                *reinterpret_cast<uint4*>(x + i0 + half) = pack8(rb);
            } else {         // k: to the cache
                store_group8<QBIT>(kv, 0, head - H, slot, i0, ra);
                store_group8<QBIT>(kv, 0, head - H, slot, i0 + half, rb);
            }
        } else {
            const int wv = w - n_rope;
            const int head = wv / (D / 8), i0 = (wv - head * (D / 8)) * 8;
            float v[8];
            if (sl.splits) slab_load8(sl, t - t0, (H + Hkv + head) * D + i0, v);
            else unpack8(*reinterpret_cast<const uint4*>(row + (int64_t)(H + Hkv + head) * D + i0), v);
            store_group8<QBIT>(kv, 1, head, slot, i0, v);
        }
    }
}

hipError_t launch_rope_kv_write(hipStream_t s, uint16_t* qkv, const float* cos_sin, const KvAddr& kv, int quant_bit,
                                int quant_group, const int64_t* seq_starts, const int64_t* start_pos,
                                const int64_t* cache_indices, int64_t max_pages, int64_t B, int64_t t0, int64_t T, int H,
                                int Hkv, int D, const SplitSlabs* qkv_slabs) {
    if (T == 0) return hipSuccess;
    SplitSlabs sl;
    if (qkv_slabs && qkv_slabs->splits > 0) {
        sl = *qkv_slabs;
        if (sl.N != (H + 2 * Hkv) * D || sl.M != T) return hipErrorInvalidValue;
    }
    if (D % 16 || (quant_bit == 8 && quant_group != 8) || (quant_bit != 0 && quant_bit != 8)) return hipErrorInvalidValue;
    const int items = (H + Hkv) * (D / 16) + Hkv * (D / 8);
    static const int forced_y = getenv("PPLHIP_ROPE_BLOCKS_PER_TOKEN") ? atoi(getenv("PPLHIP_ROPE_BLOCKS_PER_TOKEN")) : 0;
    int gy = (items + 255) / 256;            // one item per thread ...
    if (gy > 4) gy = 4;
    if (T >= 4096 && gy > 2) gy = 2;         // ... unless the launch fills the chip many times over anyway
    if (forced_y > 0) gy = forced_y;
    const dim3 grid((unsigned)T, (unsigned)gy);
    if (quant_bit == 8)
        hipLaunchKernelGGL(rope_kv_write_kernel<8>, grid, dim3(256), 0, s, qkv, cos_sin, kv, seq_starts,
                           start_pos, cache_indices, max_pages, B, t0, H, Hkv, D, sl);
    else
        hipLaunchKernelGGL(rope_kv_write_kernel<0>, grid, dim3(256), 0, s, qkv, cos_sin, kv, seq_starts,
                           start_pos, cache_indices, max_pages, B, t0, H, Hkv, D, sl);
    return hipGetLastError();
}

}  // namespace pplhip
