// K8 MultiHeadCacheAttention, decode phase: one query row per (request, head) against kv_len cached keys/values.
// THE roofline kernel of the hot path: ~1-3 flop/byte, bound by HBM bandwidth.
//
// Layout of the work (wave64, gfx950):
//   grid  = (H, nb, split)      one workgroup per (query head, request[, K-split])
//   block = NW waves            each wave owns an interleaved set of token groups
//   a KV row (D channels) is covered by LPT = D*elt/16 lanes with ONE 16-byte load each (coalesced:
//   consecutive lanes read consecutive 16-B pieces, consecutive token rows are adjacent in layout 3 / inside
//   a page), so a wave-instruction fetches TPW = 64/LPT complete rows (1 KiB).  UNROLL token groups are in
//   flight per wave before the first use (>= 8 x 16 B loads per lane outstanding: K, V, and the scales).
//   Every lane keeps q (its channels), a private online-softmax state (m, l) shared by the LPT lanes of its
//   token group, and fp32 accumulators for its channels; the TPW token groups of a wave and the NW waves of
//   a block are merged once at the end (log-sum-exp merge through shuffles, then LDS).
//   int8 KV: bytes are biased to unsigned (x ^ 0x80), converted with v_cvt_f32_ubyteN and the -128 bias is
//   folded out algebraically (sum(q) and sum(p*scale) corrections) -- the per-group fp16 scale multiplies the
//   8-channel partial dot product, not each element.
// Numerics: fp32 everywhere, output rounded to fp16 once.  Oracle: ref_attention (oracle/llama_ref.c).
#include <stdlib.h>
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

template <int QBIT, int D>
__global__ void attn_decode_kernel(const uint16_t* __restrict__ qkv, KvAddr kv, const int64_t* __restrict__ seq_starts,
                                   const int64_t* __restrict__ start_pos, const int64_t* __restrict__ cache_indices,
                                   int64_t max_pages, int H, int Hkv, int split, float* __restrict__ workspace,
                                   uint16_t* __restrict__ out) {
    using C = DecodeCfg<QBIT, D>;
    constexpr int CH = C::CH, LPT = C::LPT, TPW = C::TPW;
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [NW][D + 2]

    const int hq = blockIdx.x;
    const int64_t b = blockIdx.y;
    const int sp = blockIdx.z;
    const int hk = hq / (H / Hkv);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nw = blockDim.x >> 6;
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

    const int64_t stride = (int64_t)nw * TPW * DEC_UNROLL;
    for (int64_t t0 = tbeg + (int64_t)wave * TPW * DEC_UNROLL; t0 < tend; t0 += stride) {
        uint4 kraw[DEC_UNROLL], vraw[DEC_UNROLL];
        uint32_t ksc[DEC_UNROLL], vsc[DEC_UNROLL];
        bool valid[DEC_UNROLL];
#pragma unroll
        for (int u = 0; u < DEC_UNROLL; ++u) {
            const int64_t tok = t0 + u * TPW + g;
            valid[u] = tok < tend;
            const int64_t slot = kv_slot(kv, cache_indices, max_pages, b, valid[u] ? tok : tbeg);
            kraw[u] = *reinterpret_cast<const uint4*>(kbase + slot * row_bytes);
            vraw[u] = *reinterpret_cast<const uint4*>(vbase + slot * row_bytes);
            if constexpr (QBIT == 8) {
                if constexpr (C::NG == 2) {
                    ksc[u] = *reinterpret_cast<const uint32_t*>(ksbase + slot * kv.ssN);
                    vsc[u] = *reinterpret_cast<const uint32_t*>(vsbase + slot * kv.ssN);
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

// split-K reduce: one wave per (request, head)
template <int D>
__global__ void attn_decode_reduce_kernel(const float* __restrict__ workspace, int split, uint16_t* __restrict__ out) {
    const int64_t bh = blockIdx.x;
    const float* ws = workspace + bh * (int64_t)split * (D + 2);
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float mm = -1e30f;
        for (int s = 0; s < split; ++s) mm = fmaxf(mm, ws[s * (D + 2) + D]);
        float ll = 0.f, o = 0.f;
        for (int s = 0; s < split; ++s) {
            const float a = __expf(ws[s * (D + 2) + D] - mm);
            ll = fmaf(ws[s * (D + 2) + D + 1], a, ll);
            o = fmaf(ws[s * (D + 2) + d], a, o);
        }
        out[bh * D + d] = f2h(o / ll);
    }
}

size_t attn_decode_workspace_bytes(int64_t nb, int H, int D, int split) {
    return split > 1 ? (size_t)nb * H * split * (D + 2) * sizeof(float) : 0;
}

template <int QBIT, int D>
static hipError_t launch_decode_t(hipStream_t s, const uint16_t* qkv, const KvAddr& kv, const int64_t* seq_starts,
                                  const int64_t* start_pos, const int64_t* cache_indices, int64_t max_pages, int64_t nb,
                                  int H, int Hkv, int split, int threads, float* workspace, uint16_t* out) {
    const int nw = threads / 64;
    const size_t lds = (size_t)nw * (D + 2) * sizeof(float);
    hipLaunchKernelGGL((attn_decode_kernel<QBIT, D>), dim3(H, (unsigned)nb, split), dim3(threads), lds, s, qkv, kv,
                       seq_starts, start_pos, cache_indices, max_pages, H, Hkv, split, workspace, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (split > 1) {
        hipLaunchKernelGGL((attn_decode_reduce_kernel<D>), dim3((unsigned)(nb * H)), dim3(D < 64 ? 64 : D), 0, s,
                           workspace, split, out);
        e = hipGetLastError();
    }
    return e;
}

hipError_t launch_attn_decode(hipStream_t s, const uint16_t* qkv, const KvAddr& kv, int quant_bit,
                              const int64_t* seq_starts, const int64_t* start_pos, const int64_t* cache_indices,
                              int64_t max_pages, int64_t nb, int H, int Hkv, int D, int64_t max_kv_len, int split,
                              int threads, float* workspace, uint16_t* out) {
    if (nb == 0) return hipSuccess;
    if (threads < 64 || threads > 64 * DEC_MAX_WAVES || threads % 64) return hipErrorInvalidValue;
    if (threads < D) threads = D;  // the final merge uses one thread per channel
    if (split < 1) split = 1;
    // grouped-query models: the MFMA kernel (k_attn_prefill.hip) reads each KV row once for the whole head group
    static const bool no_gqa = getenv("PPLHIP_ATTN_NOGQA") != nullptr;
    const int grp = H / Hkv;
    if (grp >= 4 && grp <= 16 && !no_gqa) {
        hipError_t e = launch_attn_decode_gqa(s, qkv, kv, quant_bit, seq_starts, start_pos, cache_indices, max_pages, nb, H,
                                              Hkv, D, split, workspace, out);
        if (e != hipSuccess || split == 1) return e;
        const dim3 rg((unsigned)(nb * H)), rb(D < 64 ? 64 : D);
        if (D == 128) hipLaunchKernelGGL((attn_decode_reduce_kernel<128>), rg, rb, 0, s, workspace, split, out);
        else if (D == 64) hipLaunchKernelGGL((attn_decode_reduce_kernel<64>), rg, rb, 0, s, workspace, split, out);
        else hipLaunchKernelGGL((attn_decode_reduce_kernel<32>), rg, rb, 0, s, workspace, split, out);
        return hipGetLastError();
    }
#define DEC_CASE(QB, DD)                                                                                            \
    if (quant_bit == QB && D == DD)                                                                                 \
        return launch_decode_t<QB, DD>(s, qkv, kv, seq_starts, start_pos, cache_indices, max_pages, nb, H, Hkv,     \
                                       split, threads, workspace, out);
    DEC_CASE(8, 128) DEC_CASE(0, 128) DEC_CASE(8, 64) DEC_CASE(0, 64) DEC_CASE(8, 32) DEC_CASE(0, 32)
#undef DEC_CASE
    return hipErrorInvalidValue;
}

}  // namespace pplhip
