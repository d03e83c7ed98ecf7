// Device-side helpers shared by the gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdlib.h>

namespace pplhip {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int WAVE = 64;

// Tuning switches of the launch heuristics (tile shapes, ring depths, split-K counts ...): the PRODUCT reads none of them -- the values the
// measurements of profiles/ settled on are compiled in.  A tuning build (make TUNING=1: -DPPLHIP_TUNING_BUILD) reads them from the
// environment, which is what the sweep scripts under profiles/probes/ use.  (Switches the product does read -- collectives, schedules,
// the A/B switches of recent changes -- are listed in INTEGRATION.md with their defaults.)
inline int tune_int(const char* name, int dflt) {
#ifdef PPLHIP_TUNING_BUILD
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
#else
    (void)name;
    return dflt;
#endif
}
inline bool tune_set(const char* name) {
#ifdef PPLHIP_TUNING_BUILD
    return getenv(name) != nullptr;
#else
    (void)name;
    return false;
#endif
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float h2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
// fp32 -> fp16 of a COMPUTED value: the empty asm pins the fp32 value in a register first.  Without it hipcc folds a preceding
// fp32 multiply and the conversion into v_fma_mixlo_f16, which rounds the exact product once to fp16 -- not the same result as
// v_mul_f32 + v_cvt_f16_f32 (fp32 rounding, then fp16 rounding: the oracle's arithmetic) when the fp32 rounding lands on an fp16
// tie -- and it picks either form PER ELEMENT: in the GEMM epilogue 7 of the 8 row groups of a tile got the fused form for two of
// their four columns and the 8th did not, so an output row depended on where its input row sat in the batch
// (profiles/probes/gemm_position_probe*.py; caught by the permutation test of tests/test_gpu_properties.py).
__device__ __forceinline__ _Float16 to_h(float f) {
    asm("" : "+v"(f));
    return (_Float16)f;
}
__device__ __forceinline__ uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, to_h(f)); }
__device__ __forceinline__ float round_h(float f) { return (float)to_h(f); }

// 16-byte vector of 8 halfs <-> floats
// 8 int8 -> 8 fp16, exact: x ^ 0x80 is the biased byte u = x + 128; v_perm puts it under the fp16 exponent 0x64
// (= 1024 + u), minus 1152 gives x.  Two VALU ops per pair.
__device__ __forceinline__ h8 cvt_i8x8_f16(uint2 v) {
    const uint32_t w0 = v.x ^ 0x80808080u, w1 = v.y ^ 0x80808080u;
    const h2 bias = {(_Float16)1152.0f, (_Float16)1152.0f};
    const h2 a = __builtin_bit_cast(h2, __builtin_amdgcn_perm(0x64646464u, w0, 0x04010400u)) - bias;
    const h2 b = __builtin_bit_cast(h2, __builtin_amdgcn_perm(0x64646464u, w0, 0x04030402u)) - bias;
    const h2 c = __builtin_bit_cast(h2, __builtin_amdgcn_perm(0x64646464u, w1, 0x04010400u)) - bias;
    const h2 d = __builtin_bit_cast(h2, __builtin_amdgcn_perm(0x64646464u, w1, 0x04030402u)) - bias;
    return h8{a[0], a[1], b[0], b[1], c[0], c[1], d[0], d[1]};
}

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
    const h8 h = __builtin_bit_cast(h8, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)h[i];
}
__device__ __forceinline__ uint4 pack8(const float* f) {
    h8 h;
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = to_h(f[i]);
    return __builtin_bit_cast(uint4, h);
}

// Decode attention reads every K/V row exactly once per step: non-temporal loads (compile-time switch PPLHIP_KV_NT) stream them
// past the caches instead of through them -- measured +5..9 % (batch 1024 kv 512: 5.92 -> 6.23 TB/s, kv 1024: 6.24 -> 6.70 TB/s,
// profiles/attn_microbench.py; MI355X_MICROARCH.md quotes 6.4 TB/s default policy vs 6.5-6.8 nt for a streaming read)
#ifndef PPLHIP_KV_NT
#define PPLHIP_KV_NT 1
#endif
typedef uint32_t kv_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 kv_stream_load(const uint4* p) {
#if PPLHIP_KV_NT
    return __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const kv_u32x4*>(p)));
#else
    return *p;
#endif
}
__device__ __forceinline__ uint32_t kv_stream_load(const uint32_t* p) {
#if PPLHIP_KV_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

// KV slab addressing (src/engine/llm_engine.cc:118-169): element strides of (layer, k/v, head, token)
// for the four cache layouts; `d` is the innermost extent (head_dim, or head_dim/group for scales).
struct KvStrides {
    int64_t sL, sKV, sH, sN;
};
__host__ __device__ inline KvStrides kv_strides(int layout, int64_t N, int64_t L, int64_t h, int64_t d) {
    KvStrides s;
    switch (layout) {
        case 0: s.sN = L * 2 * h * d; s.sL = 2 * h * d; s.sKV = h * d; s.sH = d; break;
        case 1: s.sL = N * 2 * h * d; s.sN = 2 * h * d; s.sKV = h * d; s.sH = d; break;
        case 2: s.sL = 2 * N * h * d; s.sKV = N * h * d; s.sN = h * d; s.sH = d; break;
        default: s.sL = 2 * h * N * d; s.sKV = h * N * d; s.sH = N * d; s.sN = d; break;
    }
    return s;
}

// everything an attention / cache-write kernel needs to find K/V rows of one layer
struct KvAddr {
    void* cache;        // fp16 or int8 base of this LAYER's K plane (kv = 0); V plane = + sKV
    uint16_t* scale;    // fp16 scales, same convention
    int64_t sKV, sH, sN;       // element strides in the cache
    int64_t ssKV, ssH, ssN;    // element strides in the scale slab
    int32_t mode, page_size;   // cache_mode (0 contiguous / 1 paged)
    int32_t page_shift;        // log2(page_size) when it is a power of two (the usual 16), else -1: a 64-bit division per KV row is ~50 VALU ops
};

// KV slot of (request b, position pos): mode 0 cache_indices[b] + pos; mode 1 paged
// (src/generator/llm_generator.cc:487,553-554; llm_engine.cc:64-71)
__device__ __forceinline__ int64_t kv_slot(const KvAddr& a, const int64_t* __restrict__ cache_indices, int64_t max_pages,
                                           int64_t b, int64_t pos) {
    if (a.mode == 0) return cache_indices[b] + pos;
    if (a.page_shift >= 0) {
        const int64_t pg = pos >> a.page_shift;
        return (cache_indices[b * max_pages + pg] << a.page_shift) + (pos & (int64_t)(a.page_size - 1));
    }
    const int64_t pg = pos / a.page_size;
    return cache_indices[b * max_pages + pg] * a.page_size + (pos - pg * a.page_size);
}

}  // namespace pplhip
