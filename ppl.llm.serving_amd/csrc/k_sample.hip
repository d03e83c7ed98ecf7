// K12 apply_penalty and K13 sample_topk_topp: the sampler behind PostProcessor
// (src/common/post_processor.h:25-43; driver src/backends/cuda/post_processor.cc:121-281).  The reference
// kernels live in ppl.llm.kernel.cuda (not in the tree); semantics are fixed by DESIGN.md "sampler" and
// restated by ref_sample / ref_penalty (oracle/llama_ref.c).  HBM/L2-bound row kernels, one workgroup per row.
#include "kernels.h"

namespace pplhip {

struct ArgMax {
    float v;
    int i;
};
__device__ __forceinline__ ArgMax am_better(ArgMax a, ArgMax b) {  // larger value, ties -> lower index
    return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
__device__ __forceinline__ ArgMax wave_argmax(ArgMax a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ArgMax b;
        b.v = __shfl_xor(a.v, o, 64);
        b.i = __shfl_xor(a.i, o, 64);
        a = am_better(a, b);
    }
    return a;
}

// greedy (top_k == 1): token = first argmax of x = logits/temperature; logprob = x[token] - logsumexp(x)
__global__ __launch_bounds__(256) void sample_greedy_kernel(const float* __restrict__ logits,
                                                            const float* __restrict__ temperatures, int vocab, int stride,
                                                            int32_t* __restrict__ out_tok, float* __restrict__ out_lp) {
    __shared__ float sv[4];
    __shared__ int si[4];
    __shared__ float ss[4];
    const int b = blockIdx.x;
    const float* row = logits + (int64_t)b * stride;
    const float t = (temperatures && temperatures[b] > 0.f) ? temperatures[b] : 1.0f;
    const float invt = 1.0f / t;
    ArgMax am{-INFINITY, 0x7fffffff};
    for (int i = threadIdx.x; i < vocab; i += 256) am = am_better(am, ArgMax{row[i] * invt, i});
    am = wave_argmax(am);
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = am.v; si[threadIdx.x >> 6] = am.i; }
    __syncthreads();
    am = ArgMax{sv[0], si[0]};
    for (int w = 1; w < 4; ++w) am = am_better(am, ArgMax{sv[w], si[w]});
    float se = 0.f;
    for (int i = threadIdx.x; i < vocab; i += 256) se += __expf(row[i] * invt - am.v);
    se = wave_sum(se);
    if ((threadIdx.x & 63) == 0) ss[threadIdx.x >> 6] = se;
    __syncthreads();
    if (threadIdx.x == 0) {
        se = ss[0] + ss[1] + ss[2] + ss[3];
        out_tok[b] = am.i;
        out_lp[b] = -logf(se);  // x[token] - (max + log sum) with x[token] == max
    }
}

hipError_t launch_sample_greedy(hipStream_t s, const float* logits, const float* temperatures, int batch, int vocab,
                                int stride, int32_t* out_tok, float* out_logprob) {
    if (batch == 0) return hipSuccess;
    hipLaunchKernelGGL(sample_greedy_kernel, dim3(batch), dim3(256), 0, s, logits, temperatures, vocab, stride, out_tok,
                       out_logprob);
    return hipGetLastError();
}

// top-k / top-p: candidates = k largest x (ties: lower index first) found by k block-wide selection passes
// (the row stays L2-resident); p = softmax over candidates; keep the shortest prefix with cumulative p >= top_p
// (at least one); pick the first candidate whose cumulative mass exceeds rand * kept mass.
// top_k <= 0 (the usual "pure top-p" request): the candidates are the whole vocabulary -- p is the softmax over ALL
// logits -- and the selection passes stop as soon as the nucleus is complete (or at TOPK_MAX candidates).
// top_k > TOPK_MAX is clamped to TOPK_MAX.  A per-request parameter never fails the batch.
constexpr int TOPK_MAX = 1024;
__global__ __launch_bounds__(256) void sample_topk_topp_kernel(const float* __restrict__ logits,
                                                               const float* __restrict__ temperatures,
                                                               const float* __restrict__ top_p, const float* __restrict__ rnd,
                                                               int vocab, int stride, int top_k, float default_top_p,
                                                               int32_t* __restrict__ out_tok, float* __restrict__ out_lp) {
    __shared__ float sv[4];
    __shared__ int si[4];
    __shared__ float ss[4];
    __shared__ float cv[TOPK_MAX];
    __shared__ int ci[TOPK_MAX];
    const int b = blockIdx.x;
    const float* row = logits + (int64_t)b * stride;
    const float t = (temperatures && temperatures[b] > 0.f) ? temperatures[b] : 1.0f;
    const float invt = 1.0f / t;
    const bool full = top_k <= 0;
    int k = full ? TOPK_MAX : (top_k < TOPK_MAX ? top_k : TOPK_MAX);
    if (k > vocab) k = vocab;
    const float tp = top_p ? top_p[b] : default_top_p;
    float pv = INFINITY, mx = 0.f, se = 0.f, cum = 0.f;
    int pi = -1, n = 0;
    for (int it = 0; it < k; ++it) {
        ArgMax am{-INFINITY, 0x7fffffff};
        for (int i = threadIdx.x; i < vocab; i += 256) {
            const float x = row[i] * invt;
            if (x < pv || (x == pv && i > pi)) am = am_better(am, ArgMax{x, i});
        }
        am = wave_argmax(am);
        if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = am.v; si[threadIdx.x >> 6] = am.i; }
        __syncthreads();
        am = ArgMax{sv[0], si[0]};
        for (int w = 1; w < 4; ++w) am = am_better(am, ArgMax{sv[w], si[w]});
        if (threadIdx.x == 0) { cv[it] = am.v; ci[it] = am.i; }
        pv = am.v; pi = am.i;
        n = it + 1;
        if (it == 0) {  // the row maximum is known: total mass of the row
            mx = am.v;
            float part = 0.f;
            for (int i = threadIdx.x; i < vocab; i += 256) part += __expf(row[i] * invt - mx);
            part = wave_sum(part);
            if ((threadIdx.x & 63) == 0) ss[threadIdx.x >> 6] = part;
            __syncthreads();
            se = ss[0] + ss[1] + ss[2] + ss[3];
        }
        __syncthreads();
        if (full) {     // (block-uniform: every thread adds the same broadcast values)
            cum += __expf(am.v - mx);
            if (cum >= tp * se) break;
        }
    }
    if (threadIdx.x == 0) {
        float tot = 0.f;
        if (full) tot = se;
        else for (int i = 0; i < n; ++i) tot += expf(cv[i] - mx);
        float c1 = 0.f;
        int keep = 0;
        for (int i = 0; i < n; ++i) { c1 += expf(cv[i] - mx) / tot; keep = i + 1; if (c1 >= tp) break; }
        float ktot = 0.f;
        for (int i = 0; i < keep; ++i) ktot += expf(cv[i] - mx);
        const float target = rnd[b] * ktot;
        float c2 = 0.f;
        int sel = keep - 1;
        for (int i = 0; i < keep; ++i) { c2 += expf(cv[i] - mx); if (c2 > target) { sel = i; break; } }
        out_tok[b] = ci[sel];
        out_lp[b] = cv[sel] - (mx + logf(se));
    }
}

size_t sample_topk_workspace_bytes(int, int, int) { return 0; }

hipError_t launch_sample_topk_topp(hipStream_t s, const float* logits, const float* temperatures, const float* top_p,
                                   const float* rnd, int batch, int vocab, int stride, int top_k, float default_top_p,
                                   void*, int32_t* out_tok, float* out_logprob) {
    if (batch == 0) return hipSuccess;
    hipLaunchKernelGGL(sample_topk_topp_kernel, dim3(batch), dim3(256), 0, s, logits, temperatures, top_p, rnd, vocab,
                       stride, top_k, default_top_p, out_tok, out_logprob);
    return hipGetLastError();
}

// penalty: count map row batch_slots[b] (uint16, saturating) counts every token the request has fed the model;
// cleared on a request's first step: start_pos[b] == 0, or row b >= decoding_batches (a prefill row -- a prompt is always
// prefilled in ONE step, so this also catches a prefix-cache hit that starts at start_pos = hit > 0 in a reused batch slot;
// the cached prompt tokens themselves are never fed and stay uncounted, as in the reference).  For counted tokens: x = x > 0 ? x/rep : x*rep; x -= presence; x -= freq*count;
// finally every logit is divided by the temperature.
__global__ __launch_bounds__(256) void penalty_kernel(float* __restrict__ logits, const float* __restrict__ temperatures,
                                                      const float* __restrict__ rep, const float* __restrict__ presence,
                                                      const float* __restrict__ frequency,
                                                      const int64_t* __restrict__ batch_slots,
                                                      const int64_t* __restrict__ token_inputs,
                                                      const int64_t* __restrict__ seq_starts,
                                                      const int64_t* __restrict__ start_pos, int vocab, int stride,
                                                      int decoding_batches, uint16_t* __restrict__ count_map) {
    const int b = blockIdx.x;
    uint16_t* cm = count_map + batch_slots[b] * (int64_t)vocab;
    if (start_pos[b] == 0 || b >= decoding_batches) {
        for (int v = threadIdx.x; v < vocab; v += 256) cm[v] = 0;
    }
    __syncthreads();
    for (int64_t t = seq_starts[b] + threadIdx.x; t < seq_starts[b + 1]; t += 256) {
        const int64_t tok = token_inputs[t];
        uint32_t* word = reinterpret_cast<uint32_t*>(cm + (tok & ~(int64_t)1));  // rows are 4-byte aligned (vocab even)
        const int sh = (tok & 1) ? 16 : 0;
        uint32_t old = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (true) {
            const uint32_t cnt = (old >> sh) & 0xffffu;
            if (cnt == 0xffffu) break;
            const uint32_t nw = old + (1u << sh);
            const uint32_t prev = atomicCAS(word, old, nw);
            if (prev == old) break;
            old = prev;
        }
    }
    __threadfence_block();
    __syncthreads();
    float* row = logits + (int64_t)b * stride;
    const float t = (temperatures && temperatures[b] > 0.f) ? temperatures[b] : 1.0f;
    const float r = rep ? rep[b] : 1.0f;
    for (int v = threadIdx.x; v < vocab; v += 256) {
        float x = row[v];
        // agent-scope load: the counts were updated by L2 atomics, a plain load could hit a stale L1 line
        const uint32_t cw = __hip_atomic_load(reinterpret_cast<uint32_t*>(cm) + (v >> 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t c = (cw >> (16 * (v & 1))) & 0xffffu;
        if (c) {
            x = x > 0.f ? x / r : x * r;
            if (presence) x -= presence[b];
            if (frequency) x -= frequency[b] * (float)c;
        }
        row[v] = x / t;
    }
}

hipError_t launch_penalty(hipStream_t s, float* logits, const float* temperatures, const float* rep,
                          const float* presence, const float* frequency, const int64_t* batch_slots,
                          const int64_t* token_inputs, const int64_t* seq_starts, const int64_t* start_pos, int batch,
                          int vocab, int stride, int decoding_batches, uint16_t* count_map) {
    if (batch == 0) return hipSuccess;
    if (vocab & 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(penalty_kernel, dim3(batch), dim3(256), 0, s, logits, temperatures, rep, presence, frequency,
                       batch_slots, token_inputs, seq_starts, start_pos, vocab, stride, decoding_batches, count_map);
    return hipGetLastError();
}

}  // namespace pplhip
