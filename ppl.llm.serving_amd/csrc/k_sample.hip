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
// Round 4: 1024 threads and 16-byte loads, four of them in flight per thread (the 256-thread scalar form took 47 us for ONE row of 32000
// logits -- 2 % of a batch-1 decode step -- in 250 dependent load rounds; now ~8).  Same arithmetic: first maximum (lowest index on
// ties), then the sum of exp(x - max) in a fixed order per thread and a fixed tree across threads.
constexpr int SG_THREADS = 1024, SG_WAVES = SG_THREADS / 64;
__global__ __launch_bounds__(SG_THREADS) void sample_greedy_kernel(const float* __restrict__ logits,
                                                                   const float* __restrict__ temperatures, int vocab, int stride,
                                                                   int32_t* __restrict__ out_tok, float* __restrict__ out_lp) {
    __shared__ float sv[SG_WAVES];
    __shared__ int si[SG_WAVES];
    __shared__ float ss[SG_WAVES];
    const int b = blockIdx.x;
    const float* row = logits + (int64_t)b * stride;
    const float t = (temperatures && temperatures[b] > 0.f) ? temperatures[b] : 1.0f;
    const float invt = 1.0f / t;
    const bool vec = (stride & 3) == 0 && (reinterpret_cast<uintptr_t>(logits) & 15) == 0;
    const int nv = vec ? vocab >> 2 : 0;   // float4 chunks; the tail (and unaligned rows) element by element
    ArgMax am{-INFINITY, 0x7fffffff};
    for (int c0 = threadIdx.x; c0 < nv; c0 += 4 * SG_THREADS) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * SG_THREADS;
            v[u] = c < nv ? reinterpret_cast<const float4*>(row)[c] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = (c0 + u * SG_THREADS) * 4;
            am = am_better(am, ArgMax{v[u].x * invt, i});
            am = am_better(am, ArgMax{v[u].y * invt, i + 1});
            am = am_better(am, ArgMax{v[u].z * invt, i + 2});
            am = am_better(am, ArgMax{v[u].w * invt, i + 3});
        }
    }
    for (int i = nv * 4 + threadIdx.x; i < vocab; i += SG_THREADS) am = am_better(am, ArgMax{row[i] * invt, i});
    am = wave_argmax(am);
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = am.v; si[threadIdx.x >> 6] = am.i; }
    __syncthreads();
    am = ArgMax{sv[0], si[0]};
    for (int w = 1; w < SG_WAVES; ++w) am = am_better(am, ArgMax{sv[w], si[w]});
    float se = 0.f;
    for (int c0 = threadIdx.x; c0 < nv; c0 += 4 * SG_THREADS) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * SG_THREADS;
            v[u] = c < nv ? reinterpret_cast<const float4*>(row)[c] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            se += (__expf(v[u].x * invt - am.v) + __expf(v[u].y * invt - am.v)) + (__expf(v[u].z * invt - am.v) + __expf(v[u].w * invt - am.v));
    }
    for (int i = nv * 4 + threadIdx.x; i < vocab; i += SG_THREADS) se += __expf(row[i] * invt - am.v);
    se = wave_sum(se);
    if ((threadIdx.x & 63) == 0) ss[threadIdx.x >> 6] = se;
    __syncthreads();
    if (threadIdx.x == 0) {
        se = 0.f;
        for (int w = 0; w < SG_WAVES; ++w) se += ss[w];
        out_tok[b] = am.i;
        out_lp[b] = -logf(se);  // x[token] - (max + log sum) with x[token] == max
    }
}

hipError_t launch_sample_greedy(hipStream_t s, const float* logits, const float* temperatures, int batch, int vocab,
                                int stride, int32_t* out_tok, float* out_logprob) {
    if (batch == 0) return hipSuccess;
    hipLaunchKernelGGL(sample_greedy_kernel, dim3(batch), dim3(SG_THREADS), 0, s, logits, temperatures, vocab, stride, out_tok,
                       out_logprob);
    return hipGetLastError();
}

// top-k / top-p.  Specification (DESIGN.md "sampler", restated by ref_sample): candidates = the k largest x = logits / temperature
// (ties: lower index first), k = top_k clamped to [1, TOPK_MAX]; p = softmax over the candidates; keep the shortest prefix of the
// sorted candidates whose cumulative p reaches top_p (at least one); pick the first kept candidate whose cumulative mass exceeds
// rand * kept mass.  top_k <= 0 (the usual "pure top-p" request): the candidates are the TOPK_MAX most probable tokens and p is
// the softmax over the WHOLE vocabulary (the nucleus is cut at TOPK_MAX candidates; the oracle does the same).  A per-request
// parameter never fails the batch.
// Work per row (one 256-thread block): max + total mass (2 passes over the row, which stays L2-resident), a 4 x 8-bit radix
// selection of the k-th largest value on order-preserving keys (4 passes; + 3 passes over the indices only when equal values
// straddle the cut), one gather pass, a bitonic sort of <= 1024 candidates in LDS and block-wide scans for the two cumulative
// decisions: ~7 row passes whatever k is.  (Round 2 ran one whole-row arg-max pass PER candidate: 50 passes at top_k = 50, up
// to 1024 in pure top-p mode -- ADVICE r2.)
constexpr int TOPK_MAX = 1024;

__device__ __forceinline__ uint32_t order_key(float x) {  // larger float <=> larger key
    const uint32_t u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// block-wide inclusive scan of one int per thread (256 threads = 4 waves); `wsum` = 4 ints of LDS
__device__ __forceinline__ int block_scan_incl(int v, int* wsum) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    __syncthreads();
    if (lane == 63) wsum[w] = v;
    __syncthreads();
    for (int i = 0; i < w; ++i) v += wsum[i];
    return v;
}

// One radix-selection pass over `vocab` elements.  key(i) -> uint32, live(i) -> bool (element still matches the digits fixed so
// far).  Counts the digit `(key >> shift) & 255` of the live elements (run-length aggregated per thread: logits of one row share
// their top byte, a plain atomic per element would serialise on one LDS word), then finds the digit that holds the `need`-th
// element counting from the top (DESC) or from the bottom (ascending).  Returns the digit; `need` becomes the rank inside it and
// `*count` the number of live elements that share it.
template <bool DESC, typename KeyFn>
__device__ __forceinline__ int radix_pass(int vocab, int shift, KeyFn key_live, int& need, int* count, int* hist, int* wsum, int* sel) {
    hist[threadIdx.x] = 0;
    __syncthreads();
    int run_bin = -1, run = 0;
    for (int i = threadIdx.x; i < vocab; i += 256) {
        uint32_t k;
        if (!key_live(i, k)) continue;
        const int bin = (int)((k >> shift) & 255u);
        if (bin == run_bin) { ++run; continue; }
        if (run) atomicAdd(&hist[run_bin], run);
        run_bin = bin; run = 1;
    }
    if (run) atomicAdd(&hist[run_bin], run);
    __syncthreads();
    const int my_bin = DESC ? 255 - (int)threadIdx.x : (int)threadIdx.x;
    const int c = hist[my_bin];
    const int incl = block_scan_incl(c, wsum);
    if (incl - c < need && need <= incl) { sel[0] = my_bin; sel[1] = need - (incl - c); sel[2] = c; }
    __syncthreads();
    const int bin = sel[0];
    need = sel[1];
    *count = sel[2];
    __syncthreads();
    return bin;
}

__global__ __launch_bounds__(256) void sample_topk_topp_kernel(const float* __restrict__ logits,
                                                               const float* __restrict__ temperatures,
                                                               const float* __restrict__ top_p, const float* __restrict__ rnd,
                                                               int vocab, int stride, int top_k, float default_top_p,
                                                               int32_t* __restrict__ out_tok, float* __restrict__ out_lp) {
    __shared__ float sf[4];
    __shared__ int wsum[4], sel[4], hist[256];
    __shared__ float cv[TOPK_MAX];   // candidate values x = logit / temperature
    __shared__ float cum[TOPK_MAX];  // inclusive cumulative mass of the sorted candidates
    __shared__ int ci[TOPK_MAX];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* row = logits + (int64_t)b * stride;
    const float t = (temperatures && temperatures[b] > 0.f) ? temperatures[b] : 1.0f;
    const float invt = 1.0f / t;
    const bool full = top_k <= 0;
    int k = full ? TOPK_MAX : (top_k < TOPK_MAX ? top_k : TOPK_MAX);
    if (k > vocab) k = vocab;
    const float tp = top_p ? top_p[b] : default_top_p;

    // row maximum and total mass
    float mx = -INFINITY;
    for (int i = tid; i < vocab; i += 256) mx = fmaxf(mx, row[i] * invt);
    mx = wave_max(mx);
    if ((tid & 63) == 0) sf[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(sf[0], sf[1]), fmaxf(sf[2], sf[3]));
    __syncthreads();
    float se = 0.f;
    for (int i = tid; i < vocab; i += 256) se += __expf(row[i] * invt - mx);
    se = wave_sum(se);
    if ((tid & 63) == 0) sf[tid >> 6] = se;
    __syncthreads();
    se = sf[0] + sf[1] + sf[2] + sf[3];
    __syncthreads();

    // the k-th largest key, most significant byte first
    uint32_t prefix = 0, mask = 0;
    int need = k, n_eq = 0;
#pragma unroll 1
    for (int shift = 24; shift >= 0; shift -= 8) {
        const int bin = radix_pass<true>(vocab, shift,
            [&](int i, uint32_t& kk) { kk = order_key(row[i] * invt); return (kk & mask) == prefix; }, need, &n_eq, hist, wsum, sel);
        prefix |= (uint32_t)bin << shift;
        mask |= 255u << shift;
    }
    // `need` of the `n_eq` elements whose key equals the cut value are candidates: the ones with the lowest indices
    int idx_cut = 0x7fffffff;
    if (need < n_eq) {
        uint32_t ip = 0, im = 0;
        int dummy = 0;
#pragma unroll 1
        for (int shift = 24; shift >= 0; shift -= 8) {
            const int bin = radix_pass<false>(vocab, shift,
                [&](int i, uint32_t& kk) { kk = (uint32_t)i; return order_key(row[i] * invt) == prefix && ((uint32_t)i & im) == ip; },
                need, &dummy, hist, wsum, sel);
            ip |= (uint32_t)bin << shift;
            im |= 255u << shift;
        }
        idx_cut = (int)ip;
    }
    // gather the k candidates, pad to a power of two, sort by (value descending, index ascending)
    if (tid == 0) sel[3] = 0;
    __syncthreads();
    for (int i = tid; i < vocab; i += 256) {
        const float x = row[i] * invt;
        const uint32_t kk = order_key(x);
        if (kk > prefix || (kk == prefix && i <= idx_cut)) {
            const int slot = atomicAdd(&sel[3], 1);
            if (slot < TOPK_MAX) { cv[slot] = x; ci[slot] = i; }
        }
    }
    int n2 = 1;
    while (n2 < k) n2 <<= 1;
    __syncthreads();
    for (int i = k + tid; i < n2; i += 256) { cv[i] = -INFINITY; ci[i] = 0x7fffffff; }
    __syncthreads();
    for (int size = 2; size <= n2; size <<= 1)
        for (int str = size >> 1; str > 0; str >>= 1) {
            for (int p = tid; p < (n2 >> 1); p += 256) {
                const int lo = ((p / str) * str * 2) + (p % str), hi = lo + str;
                const bool desc = ((lo & size) == 0);  // this sub-sequence ends up "best first"
                const float va = cv[lo], vb = cv[hi];
                const int ia = ci[lo], ib = ci[hi];
                const bool a_first = va > vb || (va == vb && ia < ib);  // a ranks before b
                if (a_first != desc) { cv[lo] = vb; cv[hi] = va; ci[lo] = ib; ci[hi] = ia; }
            }
            __syncthreads();
        }
    // masses exp(x - max) of the sorted candidates and their inclusive scan (Hillis-Steele over <= 1024 entries; every round reads
    // all its operands into registers, then a barrier, then adds: no second buffer needed)
    float xv[TOPK_MAX / 256];
#pragma unroll
    for (int j = 0; j < TOPK_MAX / 256; ++j) {
        const int i = tid + j * 256;
        xv[j] = i < k ? cv[i] : -INFINITY;  // keep the candidate's x for the logprob
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TOPK_MAX / 256; ++j) {
        const int i = tid + j * 256;
        if (i < n2) cum[i] = i < k ? expf(xv[j] - mx) : 0.f;
    }
    __syncthreads();
    for (int o = 1; o < n2; o <<= 1) {
        float add[TOPK_MAX / 256];
#pragma unroll
        for (int j = 0; j < TOPK_MAX / 256; ++j) {
            const int i = tid + j * 256;
            add[j] = (i < n2 && i >= o) ? cum[i - o] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TOPK_MAX / 256; ++j) {
            const int i = tid + j * 256;
            if (i < n2) cum[i] += add[j];
        }
        __syncthreads();
    }
    // keep = 1 + first i with cum[i] / tot >= tp (all k when none); tot = whole-row mass in pure top-p mode
    const float tot = full ? se : cum[k - 1];
    if (tid == 0) { sel[0] = k - 1; sel[1] = 0x7fffffff; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TOPK_MAX / 256; ++j) {
        const int i = tid + j * 256;
        if (i < k && cum[i] / tot >= tp) atomicMin(&sel[0], i);
    }
    __syncthreads();
    const int keep = sel[0] + 1;
    const float target = rnd[b] * cum[keep - 1];
#pragma unroll
    for (int j = 0; j < TOPK_MAX / 256; ++j) {
        const int i = tid + j * 256;
        if (i < keep && cum[i] > target) atomicMin(&sel[1], i);
    }
    __syncthreads();
    const int pick = sel[1] < keep ? sel[1] : keep - 1;
#pragma unroll
    for (int j = 0; j < TOPK_MAX / 256; ++j)
        if (tid + j * 256 == pick) {
            out_tok[b] = ci[pick];
            out_lp[b] = xv[j] - (mx + logf(se));
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
