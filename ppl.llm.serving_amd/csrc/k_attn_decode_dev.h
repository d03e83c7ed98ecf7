// Device side of the decode attention kernel (K8): configuration and the workgroup body attn_decode_body; k_attn_decode.hip
// wraps it into the kernel (the body is a header so that probes can run it inside other launches).  Design notes:
// k_attn_decode.hip.
#pragma once
#include "kernels.h"

namespace pplhip {

template <int QBIT, int D>
struct DecodeCfg {
    static constexpr int ELT = QBIT == 8 ? 1 : 2;
    static constexpr int CH = 16 / ELT;       // channels per lane
    static constexpr int LPT = D / CH;        // lanes per token row
    static constexpr int TPW = 64 / LPT;      // token rows per wave-load
    static constexpr int NG = CH / 8;         // int8: quant groups per lane (group = 8 channels)
};

constexpr int DEC_UNROLL = 4;
constexpr int DEC_MAX_WAVES = 8;

// one workgroup's share: query head hq of decode request b, K-split sp of `split`; nw = waves of the block
template <int QBIT, int D>
__device__ __forceinline__ void attn_decode_body(const uint16_t* __restrict__ qkv, const KvAddr& kv, const int64_t* __restrict__ seq_starts,
                                                 const int64_t* __restrict__ start_pos, const int64_t* __restrict__ cache_indices,
                                                 int64_t max_pages, int H, int Hkv, int split, float* __restrict__ workspace,
                                                 uint16_t* __restrict__ out, int hq, int64_t b, int sp, int nw, float* smem) {
    using C = DecodeCfg<QBIT, D>;
    constexpr int CH = C::CH, LPT = C::LPT, TPW = C::TPW;
    // smem: [nw][D + 2] floats, provided by the caller

    const int hk = hq / (H / Hkv);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane / LPT;          // token group inside the wave
    const int c = lane - g * LPT;      // channel block of this lane
    const int ch0 = c * CH;

    const int64_t kv_len = start_pos[b] + 1;  // decode row: one new token at position start_pos[b]
    // token range of this split
    const int64_t per = (kv_len + split - 1) / split;
    const int64_t tbeg = sp * per;
    const int64_t tend = (tbeg + per < kv_len) ? tbeg + per : kv_len;

    // q fragment (unscaled fp16 -> fp32); the softmax scale is applied to the score
    const uint16_t* qrow = qkv + seq_starts[b] * (int64_t)(H + 2 * Hkv) * D + (int64_t)hq * D + ch0;
    float q[CH];
    if constexpr (CH == 16) {
        unpack8(*reinterpret_cast<const uint4*>(qrow), q);
        unpack8(*reinterpret_cast<const uint4*>(qrow + 8), q + 8);
    } else {
        unpack8(*reinterpret_cast<const uint4*>(qrow), q);
    }
    float qsum[C::NG > 0 ? C::NG : 1];
    if constexpr (QBIT == 8) {
#pragma unroll
        for (int gi = 0; gi < C::NG; ++gi) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) s += q[gi * 8 + i];
            qsum[gi] = s * 128.0f;
        }
    }
    const float sm_scale = 1.0f / sqrtf((float)D);

    float m = -1e30f, l = 0.f;
    float acc[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) acc[i] = 0.f;
    float vcorr[C::NG > 0 ? C::NG : 1];  // int8: sum of p*scale per group (x128 bias correction)
#pragma unroll
    for (int gi = 0; gi < (C::NG > 0 ? C::NG : 1); ++gi) vcorr[gi] = 0.f;

    const char* kbase = reinterpret_cast<const char*>(kv.cache) + ((int64_t)hk * kv.sH + ch0) * C::ELT;
    const char* vbase = kbase + kv.sKV * C::ELT;
    const uint16_t* ksbase = kv.scale + (int64_t)hk * kv.ssH + ch0 / 8;
    const uint16_t* vsbase = ksbase + kv.ssKV;
    const int64_t row_bytes = kv.sN * C::ELT;

    // Paged cache, power-of-two pages: the wave keeps 64 consecutive page ids of the request in ONE register (lane j: page pg0 + j) and
    // looks them up with ds_bpermute, instead of a dependent page-table load in front of every group of KV loads.  The window moves
    // (one dependent load) when the wave's tokens leave it: every 1024 tokens at 16-token pages.
    // (page ids as 32-bit numbers: pages of >= 4 tokens in a slab that fits 288 GB stay far below 2^31)
    const bool lane_pages = kv.mode == 1 && kv.page_shift >= 2;
    const int64_t* const page_row = cache_indices + b * max_pages;
    int64_t pg0 = 0;
    int page_reg = 0;
    const int64_t stride = (int64_t)nw * TPW * DEC_UNROLL;
    const int64_t tfirst = tbeg + (int64_t)wave * TPW * DEC_UNROLL;
    if (lane_pages && tfirst < tend) {
        pg0 = tfirst >> kv.page_shift;
        const int64_t pj = pg0 + lane;
        page_reg = (int)page_row[pj < max_pages ? pj : max_pages - 1];
    }
    for (int64_t t0 = tfirst; t0 < tend; t0 += stride) {
        uint4 kraw[DEC_UNROLL], vraw[DEC_UNROLL];
        uint32_t ksc[DEC_UNROLL], vsc[DEC_UNROLL];
        bool valid[DEC_UNROLL];
        if (lane_pages && ((t0 + TPW * DEC_UNROLL - 1) >> kv.page_shift) - pg0 > 63) {  // wave-uniform
            pg0 = t0 >> kv.page_shift;
            const int64_t pj = pg0 + lane;
            page_reg = (int)page_row[pj < max_pages ? pj : max_pages - 1];
        }
#pragma unroll
        for (int u = 0; u < DEC_UNROLL; ++u) {
            const int64_t tok = t0 + u * TPW + g;
            valid[u] = tok < tend;
            int64_t slot;
            if (lane_pages) {
                const int64_t tk = valid[u] ? tok : t0;
                const int pid = __shfl(page_reg, (int)((tk >> kv.page_shift) - pg0), 64);
                slot = ((int64_t)pid << kv.page_shift) + (tk & (int64_t)(kv.page_size - 1));
            } else {
                slot = kv_slot(kv, cache_indices, max_pages, b, valid[u] ? tok : tbeg);
            }
            kraw[u] = kv_stream_load(reinterpret_cast<const uint4*>(kbase + slot * row_bytes));
            vraw[u] = kv_stream_load(reinterpret_cast<const uint4*>(vbase + slot * row_bytes));
            if constexpr (QBIT == 8) {
                if constexpr (C::NG == 2) {
                    ksc[u] = kv_stream_load(reinterpret_cast<const uint32_t*>(ksbase + slot * kv.ssN));
                    vsc[u] = kv_stream_load(reinterpret_cast<const uint32_t*>(vsbase + slot * kv.ssN));
                } else {
                    ksc[u] = ksbase[slot * kv.ssN];
                    vsc[u] = vsbase[slot * kv.ssN];
                }
            }
        }
        float s[DEC_UNROLL];
#pragma unroll
        for (int u = 0; u < DEC_UNROLL; ++u) {
            float d = 0.f;
            if constexpr (QBIT == 8) {
                const uint32_t w[4] = {kraw[u].x ^ 0x80808080u, kraw[u].y ^ 0x80808080u, kraw[u].z ^ 0x80808080u,
                                       kraw[u].w ^ 0x80808080u};
#pragma unroll
                for (int gi = 0; gi < C::NG; ++gi) {
                    float pd = -qsum[gi];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const uint32_t word = w[(gi * 8 + i) >> 2];
                        const float kf = (float)((word >> (8 * (i & 3))) & 0xffu);
                        pd = fmaf(q[gi * 8 + i], kf, pd);
                    }
                    const float sc = h2f((uint16_t)(ksc[u] >> (16 * gi)));
                    d = fmaf(pd, sc, d);
                }
            } else {
                float kf[8];
                unpack8(kraw[u], kf);
#pragma unroll
                for (int i = 0; i < 8; ++i) d = fmaf(q[i], kf[i], d);
            }
            // reduce over the LPT lanes of the token row
#pragma unroll
            for (int o = 1; o < LPT; o <<= 1) d += __shfl_xor(d, o, 64);
            s[u] = valid[u] ? d * sm_scale : -1e30f;
        }
        float mnew = m;
#pragma unroll
        for (int u = 0; u < DEC_UNROLL; ++u) mnew = fmaxf(mnew, s[u]);
        const float alpha = __expf(m - mnew);
        m = mnew;
        l *= alpha;
#pragma unroll
        for (int i = 0; i < CH; ++i) acc[i] *= alpha;
        if constexpr (QBIT == 8) {
#pragma unroll
            for (int gi = 0; gi < C::NG; ++gi) vcorr[gi] *= alpha;
        }
#pragma unroll
        for (int u = 0; u < DEC_UNROLL; ++u) {
            const float p = valid[u] ? __expf(s[u] - m) : 0.f;
            l += p;
            if constexpr (QBIT == 8) {
                const uint32_t w[4] = {vraw[u].x ^ 0x80808080u, vraw[u].y ^ 0x80808080u, vraw[u].z ^ 0x80808080u,
                                       vraw[u].w ^ 0x80808080u};
#pragma unroll
                for (int gi = 0; gi < C::NG; ++gi) {
                    const float ps = p * h2f((uint16_t)(vsc[u] >> (16 * gi)));
                    vcorr[gi] += ps;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const uint32_t word = w[(gi * 8 + i) >> 2];
                        const float vf = (float)((word >> (8 * (i & 3))) & 0xffu);
                        acc[gi * 8 + i] = fmaf(ps, vf, acc[gi * 8 + i]);
                    }
                }
            } else {
                float vf[8];
                unpack8(vraw[u], vf);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = fmaf(p, vf[i], acc[i]);
            }
        }
    }
    if constexpr (QBIT == 8) {
#pragma unroll
        for (int gi = 0; gi < C::NG; ++gi)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[gi * 8 + i] = fmaf(-128.0f, vcorr[gi], acc[gi * 8 + i]);
    }

    // merge the TPW token groups of the wave (lanes with equal channel block c)
#pragma unroll
    for (int o = LPT; o < 64; o <<= 1) {
        const float mo = __shfl_xor(m, o, 64);
        const float lo = __shfl_xor(l, o, 64);
        const float mn = fmaxf(m, mo);
        const float a0 = __expf(m - mn), a1 = __expf(mo - mn);
        l = l * a0 + lo * a1;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const float ao = __shfl_xor(acc[i], o, 64);
            acc[i] = acc[i] * a0 + ao * a1;
        }
        m = mn;
    }
    // merge the waves through LDS
    float* my = smem + wave * (D + 2);
    if (g == 0) {
#pragma unroll
        for (int i = 0; i < CH; ++i) my[ch0 + i] = acc[i];
        if (c == 0) { my[D] = m; my[D + 1] = l; }
    }
    __syncthreads();
    if (threadIdx.x < D) {
        const int d = threadIdx.x;
        float mm = -1e30f;
        for (int w = 0; w < nw; ++w) mm = fmaxf(mm, smem[w * (D + 2) + D]);
        float ll = 0.f, o = 0.f;
        for (int w = 0; w < nw; ++w) {
            const float a = __expf(smem[w * (D + 2) + D] - mm);
            ll = fmaf(smem[w * (D + 2) + D + 1], a, ll);
            o = fmaf(smem[w * (D + 2) + d], a, o);
        }
        if (split == 1) {
            out[(b * H + hq) * (int64_t)D + d] = f2h(o / ll);
        } else {
            float* ws = workspace + ((b * H + hq) * (int64_t)split + sp) * (D + 2);
            ws[d] = o;
            if (d == 0) { ws[D] = mm; ws[D + 1] = ll; }
        }
    }
}


}  // namespace pplhip
