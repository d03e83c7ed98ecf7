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
#include <hip/hip_ext.h>
#include "k_attn_decode_dev.h"

namespace pplhip {

template <int QBIT, int D>
__global__ void attn_decode_kernel(const uint16_t* __restrict__ qkv, KvAddr kv, const int64_t* __restrict__ seq_starts,
                                   const int64_t* __restrict__ start_pos, const int64_t* __restrict__ cache_indices,
                                   int64_t max_pages, int H, int Hkv, int split, float* __restrict__ workspace,
                                   uint16_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [NW][D + 2]
    attn_decode_body<QBIT, D>(qkv, kv, seq_starts, start_pos, cache_indices, max_pages, H, Hkv, split, workspace, out, (int)blockIdx.x,
                              (int64_t)blockIdx.y, (int)blockIdx.z, (int)(blockDim.x >> 6), smem);
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
                                  int H, int Hkv, int split, int threads, float* workspace, uint16_t* out, hipEvent_t t0,
                                  hipEvent_t t1) {
    const int nw = threads / 64;
    const size_t lds = (size_t)nw * (D + 2) * sizeof(float);
    // t0 / t1: start and stop timestamps taken from the kernel's own dispatch packet -- no extra barrier packets on the stream
    // (an hipEventRecord pair around the launch costs ~10 us of GPU time)
    if (t0 && t1)
        hipExtLaunchKernelGGL((attn_decode_kernel<QBIT, D>), dim3(H, (unsigned)nb, split), dim3(threads), lds, s, t0, t1, 0, qkv, kv,
                              seq_starts, start_pos, cache_indices, max_pages, H, Hkv, split, workspace, out);
    else
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
                              int threads, float* workspace, uint16_t* out, hipEvent_t t0, hipEvent_t t1) {
    if (nb == 0) return hipSuccess;
    static const int forced_tpb = tune_int("PPLHIP_ATTN_TPB", 0);  // tuning only
    if (forced_tpb) threads = forced_tpb;
    if (threads < 64 || threads > 64 * DEC_MAX_WAVES || threads % 64) return hipErrorInvalidValue;
    if (threads < D) threads = D;  // the final merge uses one thread per channel
    if (split < 1) split = 1;
    // grouped-query models: the MFMA kernel (k_attn_prefill.hip) reads each KV row once for the whole head group
    static const bool no_gqa = tune_set("PPLHIP_ATTN_NOGQA");
    if (attn_decode_gqa_supported(quant_bit, H, Hkv, D) && !no_gqa) {
        hipError_t e = launch_attn_decode_gqa(s, qkv, kv, quant_bit, seq_starts, start_pos, cache_indices, max_pages, nb, H,
                                              Hkv, D, split, workspace, out, t0, t1);
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
                                       split, threads, workspace, out, t0, t1);
    DEC_CASE(8, 128) DEC_CASE(0, 128) DEC_CASE(8, 64) DEC_CASE(0, 64) DEC_CASE(8, 32) DEC_CASE(0, 32)
#undef DEC_CASE
    return hipErrorInvalidValue;
}

}  // namespace pplhip
