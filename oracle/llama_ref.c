/*
 * oracle/llama_ref.c -- CPU restatement of the hot path of ppl.llm.serving (TEST INFRASTRUCTURE ONLY).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / reported baseline -- never as the thing shipped.  The product (libpplhip.so) does not
 * link, load or call anything in this directory.
 *
 * PARITY UNPINNED for the model arithmetic: the reference executes the transformer inside ppl.nn /
 * ppl.llm.kernel.cuda, fetched at configure time from github.com/OpenPPL/ppl.nn @ master (unpinned,
 * /root/reference/cmake/deps.cmake:92-106); neither is in the reference tree, the reference has no CPU
 * backend (cmake/llm.cmake:10-17) and no numeric test (test/test_prefix_cache_mgr.cc:25-66 prints only).
 * What IS pinned:
 *   - the runtime contract this file implements: the 11 inputs / 1 output bound by index in
 *     src/engine/llm_engine.h:124-138, written per step by src/engine/llm_engine.cc:29-111, with the KV
 *     slab shapes of src/engine/llm_engine.cc:118-169 and the packing of
 *     src/generator/llm_generator.cc:263-298;
 *   - the model arithmetic against an INDEPENDENT oracle, HuggingFace transformers' LlamaForCausalLM
 *     (fp32, CPU), through the committed fixtures tests/golden/hf_tiny_*.npz (generator:
 *     oracle/make_hf_golden.py);
 *   - the host logic (hashing, prefix cache, packing) against values captured from the reference's own
 *     code, tests/golden/host_logic.json (see oracle/host_logic.py).
 * Where the reference is silent the numerics are fixed by DESIGN.md "numerics" and restated here.
 *
 * Everything is plain C (gcc -O3 -fopenmp -march=x86-64-v3); fp16 storage is emulated by rounding floats
 * through IEEE binary16 with F16C.
 */
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define REF_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------------
 * fp16 helpers
 * ---------------------------------------------------------------------------------------------- */
typedef uint16_t f16;
static inline float h2f(f16 h) { return _cvtsh_ss(h); }
static inline f16 f2h(float f) { return _cvtss_sh(f, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC); }
/* arithmetic mode of the whole library (ref_set_mode; default 0 = the specification the device is held against):
 *   bit 0  REF_MODE_FP32_ACT  no fp16 rounding of computed activations, an unquantised KV slab holds fp32: the ALGORITHM in
 *                             (almost) exact arithmetic -- pinned against HuggingFace fp32 at ~1e-5 (tests/test_oracle_hf.py);
 *                             separates "algorithm wrong" from "rounding differs"
 *   bit 1  REF_MODE_F64_ACC   dot products accumulate in double (an exactly rounded fp32 result)
 *   bit 2  REF_MODE_ALT_ORDER dot products in a different, equally valid fp32 summation order: oracle(mode 4) vs oracle(mode 0)
 *                             measures how far two correct fp16 implementations of this specification drift apart
 *                             (the noise floor the device tolerances are derived from, tests/test_gpu_fulldepth.py) */
/*   bit 3  REF_MODE_ALT_ORDER2 a third valid order (one 8-lane accumulator over 8-element steps, lanes summed as a tree): with
 *                             three orders the noise floor is the largest of three pairwise distances instead of one sample */
enum { REF_MODE_FP32_ACT = 1, REF_MODE_F64_ACC = 2, REF_MODE_ALT_ORDER = 4, REF_MODE_ALT_ORDER2 = 8 };
static int g_mode = 0;
static inline float rh(float f) { return (g_mode & REF_MODE_FP32_ACT) ? f : h2f(f2h(f)); } /* round through fp16 */

/* ------------------------------------------------------------------------------------------------
 * model description -- field-for-field the same as pplhip_model_desc (include/pplhip.h), which carries
 * params.json (src/common/config.cc:31-148) plus what the exported graph encodes.
 * ---------------------------------------------------------------------------------------------- */
typedef struct ref_model_desc {
    int32_t hidden_dim, intermediate_dim, num_layers, num_heads, num_kv_heads, vocab_size;
    float norm_eps, rope_theta;
    int32_t max_position;
    int32_t cache_quant_bit, cache_quant_group, cache_layout, cache_mode, page_size;
    int32_t weight_quant_bit, weight_quant_group;
    int32_t act_quant_bit; /* 8 = online_i8i8 (W8A8): activations quantised per token in front of every layer linear */
} ref_model_desc;

/* one step: ModelInput as the runtime sees it (src/engine/llm_engine.h:40-60, llm_engine.cc:29-111) */
typedef struct ref_step {
    int64_t batch, num_tokens, decoding_batches, max_seq_len, max_kv_len, max_pages;
    const int64_t* token_inputs;
    const int64_t* seq_starts;
    const int64_t* kv_starts;
    const int64_t* start_pos;
    const int64_t* cache_indices;
    int32_t req_list_changed;
} ref_step;

typedef struct ref_linear {
    int32_t N, K;      /* y[.,N] = x[.,K] W^T */
    int32_t qbit;      /* 0 fp16, 8, 4 */
    int32_t group;     /* W4 group */
    f16* w16;          /* qbit 0: [N,K] */
    int8_t* w8;        /* qbit 8: [N,K] */
    uint8_t* w4;       /* qbit 4: [N,K/2], low nibble = even k, value = nibble - 8 */
    f16* scale;        /* qbit 8: [N]; qbit 4: [N, K/group] */
    int32_t a8;        /* qbit 8 only: int8 activations (online_i8i8) */
} ref_linear;

typedef struct ref_layer {
    f16* attn_norm;
    ref_linear wqkv, wo, w13, w2;
    f16* ffn_norm;
} ref_layer;

typedef struct ref_model {
    ref_model_desc d;
    int32_t tp_size, tp_rank;
    int32_t H, Hkv, D, inter; /* per-rank */
    int32_t vocab_local;
    f16* embed;       /* [vocab, hidden] replicated */
    ref_layer* layers;
    f16* norm;
    ref_linear output;  /* fp16 [vocab/tp, hidden] */
    float* rope;      /* [max_position, D]: cos[0..D/2) then sin[0..D/2) */
    /* KV slab */
    uint64_t kv_tokens;
    void* kv_cache;   /* f16 (fp32 when kv_f32) or int8 */
    int32_t kv_f32;
    f16* kv_scale;
} ref_model;

/* ------------------------------------------------------------------------------------------------
 * synthetic weights: counter-based generator shared bit-for-bit with csrc/synth.hip
 * ---------------------------------------------------------------------------------------------- */
static inline uint64_t mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31; return z;
}
static inline uint64_t synth_key(uint64_t seed, uint32_t tensor_id, uint32_t stream) {
    return mix64(seed ^ ((uint64_t)tensor_id * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)stream * 0xD1B54A32D192ED03ull));
}
static inline uint64_t synth_val(uint64_t key, uint64_t idx) { return mix64(key + idx * 0x9E3779B97F4A7C15ull); }
static inline float synth_unit(uint64_t v) { return (float)(uint32_t)(v >> 40) * (1.0f / 16777216.0f); } /* [0,1) */

enum { SYN_F16_SYM = 0, SYN_I8 = 1, SYN_I4 = 2, SYN_SCALE = 3, SYN_NORM = 4 };
#define SYN_AMP 0.034641016f /* 0.02*sqrt(3): uniform with std 0.02 */

/* kind: SYN_F16_SYM -> fp16 uniform(-amp,amp); SYN_I8 -> int8 uniform[-127,127]; SYN_I4 -> packed
 * nibbles (n = number of BYTES, 2 values each); SYN_SCALE -> fp16 amp*(0.5+u); SYN_NORM -> fp16 1+0.1(u-.5) */
REF_API void ref_synth_fill(int kind, uint64_t seed, uint32_t tensor_id, uint32_t stream, float amp, uint64_t n,
                            void* out) {
    const uint64_t key = synth_key(seed, tensor_id, stream);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        switch (kind) {
            case SYN_F16_SYM: ((f16*)out)[i] = f2h((synth_unit(synth_val(key, i)) - 0.5f) * 2.0f * amp); break;
            case SYN_I8: ((int8_t*)out)[i] = (int8_t)((int)((uint32_t)(synth_val(key, i) >> 32) % 255u) - 127); break;
            case SYN_I4: {
                uint32_t lo = (uint32_t)(synth_val(key, 2 * i) >> 32) & 15u;
                uint32_t hi = (uint32_t)(synth_val(key, 2 * i + 1) >> 32) & 15u;
                ((uint8_t*)out)[i] = (uint8_t)(lo | (hi << 4));
            } break;
            case SYN_SCALE: ((f16*)out)[i] = f2h(amp * (0.5f + synth_unit(synth_val(key, i)))); break;
            case SYN_NORM: ((f16*)out)[i] = f2h(1.0f + 0.1f * (synth_unit(synth_val(key, i)) - 0.5f)); break;
        }
    }
}

/* tensor ids: (layer+1)*32 + slot, layer = -1 for globals */
enum { T_EMBED = 0, T_ATTN_NORM = 1, T_WQKV = 2, T_WQKV_S = 3, T_WO = 4, T_WO_S = 5, T_FFN_NORM = 6,
       T_W13 = 7, T_W13_S = 8, T_W2 = 9, T_W2_S = 10, T_NORM = 11, T_OUTPUT = 12 };
static inline uint32_t tid(int layer, int slot) { return (uint32_t)((layer + 1) * 32 + slot); }

/* ------------------------------------------------------------------------------------------------
 * construction
 * ---------------------------------------------------------------------------------------------- */
static void linear_alloc(ref_linear* l, int N, int K, int qbit, int group) {
    memset(l, 0, sizeof(*l));
    l->N = N; l->K = K; l->qbit = qbit; l->group = group;
    if (qbit == 0) l->w16 = (f16*)calloc((size_t)N * K, 2);
    else if (qbit == 8) { l->w8 = (int8_t*)calloc((size_t)N * K, 1); l->scale = (f16*)calloc(N, 2); }
    else { l->w4 = (uint8_t*)calloc((size_t)N * K / 2, 1); l->scale = (f16*)calloc((size_t)N * (K / group), 2); }
}
static void linear_free(ref_linear* l) { free(l->w16); free(l->w8); free(l->w4); free(l->scale); }

/* fp32 cos/sin table computed in double -- identical code in csrc (pplhip_build_rope_table). */
REF_API void ref_build_rope_table(float* out, int32_t max_position, int32_t head_dim, float theta) {
    const int half = head_dim / 2;
    for (int p = 0; p < max_position; ++p)
        for (int i = 0; i < half; ++i) {
            double freq = pow((double)theta, -2.0 * (double)i / (double)head_dim);
            double a = (double)p * freq;
            out[(size_t)p * head_dim + i] = (float)cos(a);
            out[(size_t)p * head_dim + half + i] = (float)sin(a);
        }
}

REF_API ref_model* ref_create(const ref_model_desc* d, int tp_size, int tp_rank) {
    ref_model* m = (ref_model*)calloc(1, sizeof(ref_model));
    m->d = *d; m->tp_size = tp_size; m->tp_rank = tp_rank;
    m->D = d->hidden_dim / d->num_heads;
    m->H = d->num_heads / tp_size;
    m->Hkv = d->num_kv_heads / tp_size;
    m->inter = d->intermediate_dim / tp_size;
    m->vocab_local = d->vocab_size / tp_size;
    const int hd = d->hidden_dim, q = d->weight_quant_bit, g = d->weight_quant_group;
    m->embed = (f16*)calloc((size_t)d->vocab_size * hd, 2);
    m->norm = (f16*)calloc(hd, 2);
    m->layers = (ref_layer*)calloc(d->num_layers, sizeof(ref_layer));
    for (int l = 0; l < d->num_layers; ++l) {
        ref_layer* L = &m->layers[l];
        L->attn_norm = (f16*)calloc(hd, 2);
        L->ffn_norm = (f16*)calloc(hd, 2);
        linear_alloc(&L->wqkv, (m->H + 2 * m->Hkv) * m->D, hd, q, g);
        linear_alloc(&L->wo, hd, m->H * m->D, q, g);
        linear_alloc(&L->w13, 2 * m->inter, hd, q, g);
        linear_alloc(&L->w2, hd, m->inter, q, g);
        if (d->act_quant_bit == 8 && q == 8) L->wqkv.a8 = L->wo.a8 = L->w13.a8 = L->w2.a8 = 1;
    }
    linear_alloc(&m->output, m->vocab_local, hd, 0, 0);
    m->rope = (float*)malloc((size_t)d->max_position * m->D * sizeof(float));
    ref_build_rope_table(m->rope, d->max_position, m->D, d->rope_theta);
    return m;
}

REF_API void ref_destroy(ref_model* m) {
    if (!m) return;
    for (int l = 0; l < m->d.num_layers; ++l) {
        ref_layer* L = &m->layers[l];
        free(L->attn_norm); free(L->ffn_norm);
        linear_free(&L->wqkv); linear_free(&L->wo); linear_free(&L->w13); linear_free(&L->w2);
    }
    linear_free(&m->output);
    free(m->layers); free(m->embed); free(m->norm); free(m->rope); free(m->kv_cache); free(m->kv_scale);
    free(m);
}

/* name -> buffer.  Names are the weight-container names of DESIGN.md. */
static int find_tensor(ref_model* m, const char* name, void** ptr, uint64_t* bytes) {
    const int hd = m->d.hidden_dim;
    if (!strcmp(name, "tok_embeddings.weight")) { *ptr = m->embed; *bytes = (uint64_t)m->d.vocab_size * hd * 2; return 0; }
    if (!strcmp(name, "norm.weight")) { *ptr = m->norm; *bytes = (uint64_t)hd * 2; return 0; }
    if (!strcmp(name, "output.weight")) { *ptr = m->output.w16; *bytes = (uint64_t)m->vocab_local * hd * 2; return 0; }
    int l = -1; char rest[128];
    if (sscanf(name, "layers.%d.%127s", &l, rest) != 2 || l < 0 || l >= m->d.num_layers) return -1;
    ref_layer* L = &m->layers[l];
    if (!strcmp(rest, "attention_norm.weight")) { *ptr = L->attn_norm; *bytes = (uint64_t)hd * 2; return 0; }
    if (!strcmp(rest, "ffn_norm.weight")) { *ptr = L->ffn_norm; *bytes = (uint64_t)hd * 2; return 0; }
    struct { const char* n; ref_linear* lin; } tab[] = {
        {"attention.wqkv", &L->wqkv}, {"attention.wo", &L->wo}, {"feed_forward.w13", &L->w13}, {"feed_forward.w2", &L->w2}};
    for (int i = 0; i < 4; ++i) {
        size_t nl = strlen(tab[i].n);
        if (strncmp(rest, tab[i].n, nl)) continue;
        ref_linear* lin = tab[i].lin;
        if (!strcmp(rest + nl, ".weight")) {
            if (lin->qbit == 0) { *ptr = lin->w16; *bytes = (uint64_t)lin->N * lin->K * 2; }
            else if (lin->qbit == 8) { *ptr = lin->w8; *bytes = (uint64_t)lin->N * lin->K; }
            else { *ptr = lin->w4; *bytes = (uint64_t)lin->N * lin->K / 2; }
            return 0;
        }
        if (!strcmp(rest + nl, ".scale") && lin->qbit) {
            *ptr = lin->scale;
            *bytes = lin->qbit == 8 ? (uint64_t)lin->N * 2 : (uint64_t)lin->N * (lin->K / lin->group) * 2;
            return 0;
        }
    }
    return -1;
}

REF_API void ref_quant_weight_rows(const f16* w, int N, int K, int8_t* q, f16* scale);

static ref_linear* find_w8_linear(ref_model* m, const char* name) {
    int l = -1; char rest[128];
    if (sscanf(name, "layers.%d.%127s", &l, rest) != 2 || l < 0 || l >= m->d.num_layers) return NULL;
    ref_layer* L = &m->layers[l];
    ref_linear* lin = !strcmp(rest, "attention.wqkv.weight") ? &L->wqkv : !strcmp(rest, "attention.wo.weight") ? &L->wo
                    : !strcmp(rest, "feed_forward.w13.weight") ? &L->w13 : !strcmp(rest, "feed_forward.w2.weight") ? &L->w2 : NULL;
    return lin && lin->qbit == 8 ? lin : NULL;
}

REF_API int ref_set_tensor(ref_model* m, const char* name, const void* data, uint64_t bytes) {
    void* p; uint64_t b;
    if (find_tensor(m, name, &p, &b)) return -6;
    /* "online" quantisation: an fp16 matrix handed to an int8 linear is quantised per output row on the way in */
    ref_linear* lin = find_w8_linear(m, name);
    if (lin && bytes == (uint64_t)lin->N * lin->K * 2) {
        ref_quant_weight_rows((const f16*)data, lin->N, lin->K, lin->w8, lin->scale);
        return 0;
    }
    if (b != bytes) return -2;
    memcpy(p, data, bytes);
    return 0;
}
REF_API int ref_get_tensor(ref_model* m, const char* name, void* data, uint64_t bytes) {
    void* p; uint64_t b;
    if (find_tensor(m, name, &p, &b)) return -6;
    if (b != bytes) return -2;
    memcpy(data, p, bytes);
    return 0;
}
REF_API int64_t ref_tensor_bytes(ref_model* m, const char* name) {
    void* p; uint64_t b;
    if (find_tensor(m, name, &p, &b)) return -6;
    return (int64_t)b;
}

static void linear_synth(ref_linear* l, uint64_t seed, int layer, int wslot, uint32_t stream) {
    if (l->qbit == 0) ref_synth_fill(SYN_F16_SYM, seed, tid(layer, wslot), stream, SYN_AMP, (uint64_t)l->N * l->K, l->w16);
    else if (l->qbit == 8) {
        ref_synth_fill(SYN_I8, seed, tid(layer, wslot), stream, 0, (uint64_t)l->N * l->K, l->w8);
        ref_synth_fill(SYN_SCALE, seed, tid(layer, wslot + 1), stream, SYN_AMP / 127.0f, l->N, l->scale);
    } else {
        ref_synth_fill(SYN_I4, seed, tid(layer, wslot), stream, 0, (uint64_t)l->N * l->K / 2, l->w4);
        ref_synth_fill(SYN_SCALE, seed, tid(layer, wslot + 1), stream, SYN_AMP / 7.0f, (uint64_t)l->N * (l->K / l->group), l->scale);
    }
}

/* replicated tensors use stream 0 on every rank; sharded ones stream = 1 + tp_rank */
REF_API int ref_init_synthetic(ref_model* m, uint64_t seed) {
    const int hd = m->d.hidden_dim;
    const uint32_t st = 1u + (uint32_t)m->tp_rank;
    ref_synth_fill(SYN_F16_SYM, seed, tid(-1, T_EMBED), 0, 1.0f, (uint64_t)m->d.vocab_size * hd, m->embed);
    ref_synth_fill(SYN_NORM, seed, tid(-1, T_NORM), 0, 0, hd, m->norm);
    ref_synth_fill(SYN_F16_SYM, seed, tid(-1, T_OUTPUT), st, SYN_AMP, (uint64_t)m->vocab_local * hd, m->output.w16);
    for (int l = 0; l < m->d.num_layers; ++l) {
        ref_layer* L = &m->layers[l];
        ref_synth_fill(SYN_NORM, seed, tid(l, T_ATTN_NORM), 0, 0, hd, L->attn_norm);
        ref_synth_fill(SYN_NORM, seed, tid(l, T_FFN_NORM), 0, 0, hd, L->ffn_norm);
        linear_synth(&L->wqkv, seed, l, T_WQKV, st);
        linear_synth(&L->wo, seed, l, T_WO, st);
        linear_synth(&L->w13, seed, l, T_W13, st);
        linear_synth(&L->w2, seed, l, T_W2, st);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * KV slab.  Shapes: src/engine/llm_engine.cc:118-169 (N = kv_cache_max_tokens, h = Hkv/TP, d = Dh):
 *   layout 0 [N,L,2,h,d] . 1 [L,N,2,h,d] . 2 [L,2,N,h,d] . 3 [L,2,h,N,d]; scale: last dim d/group.
 * Bytes per token: src/backends/cuda/resource_manager.cc:381-387.
 * ---------------------------------------------------------------------------------------------- */
typedef struct kv_strides { int64_t sL, sKV, sH, sN; } kv_strides;
static kv_strides kv_strides_of(int layout, int64_t N, int64_t L, int64_t h, int64_t d) {
    kv_strides s;
    switch (layout) {
        case 0: s.sN = L * 2 * h * d; s.sL = 2 * h * d; s.sKV = h * d; s.sH = d; break;
        case 1: s.sL = N * 2 * h * d; s.sN = 2 * h * d; s.sKV = h * d; s.sH = d; break;
        case 2: s.sL = 2 * N * h * d; s.sKV = N * h * d; s.sN = h * d; s.sH = d; break;
        default: s.sL = 2 * h * N * d; s.sKV = h * N * d; s.sH = N * d; s.sN = d; break;
    }
    return s;
}

/* an unquantised slab holds fp16, or fp32 in REF_MODE_FP32_ACT (the mode in force when the slab is allocated) */
REF_API int ref_kv_alloc(ref_model* m, uint64_t tokens) {
    free(m->kv_cache); free(m->kv_scale); m->kv_scale = NULL;
    m->kv_tokens = tokens;
    const uint64_t elems = tokens * m->d.num_layers * 2 * m->Hkv * m->D;
    m->kv_f32 = m->d.cache_quant_bit == 0 && (g_mode & REF_MODE_FP32_ACT);
    m->kv_cache = calloc(elems, m->d.cache_quant_bit == 8 ? 1 : (m->kv_f32 ? 4 : 2));
    if (m->d.cache_quant_bit == 8) m->kv_scale = (f16*)calloc(elems / m->d.cache_quant_group, 2);
    return m->kv_cache ? 0 : -3;
}
REF_API void* ref_kv_ptr(ref_model* m, int which) { return which ? (void*)m->kv_scale : m->kv_cache; }
REF_API uint64_t ref_kv_bytes(ref_model* m, int which) {
    const uint64_t elems = m->kv_tokens * m->d.num_layers * 2 * m->Hkv * m->D;
    if (which) return m->d.cache_quant_bit == 8 ? elems / m->d.cache_quant_group * 2 : 0;
    return elems * (m->d.cache_quant_bit == 8 ? 1 : (m->kv_f32 ? 4 : 2));
}
REF_API void ref_set_mode(int mode) { g_mode = mode; }
REF_API int ref_get_mode(void) { return g_mode; }

/* KV slot of (request b, absolute position pos): mode 0 cache_indices[b] + pos
 * (src/generator/llm_generator.cc:487, llm_engine.cc:64-66); mode 1 page_list[b, pos/P]*P + pos%P
 * (llm_generator.cc:553-554, 278-296; SURVEY.md section 10 PageManager). */
static inline int64_t kv_slot(const ref_model_desc* d, const int64_t* cache_indices, int64_t max_pages, int64_t b, int64_t pos) {
    if (d->cache_mode == 0) return cache_indices[b] + pos;
    return cache_indices[b * max_pages + pos / d->page_size] * d->page_size + pos % d->page_size;
}

/* ------------------------------------------------------------------------------------------------
 * operators (DESIGN.md "numerics").  Activations are float arrays holding fp16-representable values.
 * ---------------------------------------------------------------------------------------------- */

/* K1 embedding gather: h[t,:] = E[token_ids[t],:] */
REF_API void ref_embedding(const int64_t* token_ids, const f16* table, int64_t T, int hidden, float* out) {
#pragma omp parallel for
    for (int64_t t = 0; t < T; ++t)
        for (int i = 0; i < hidden; ++i) out[t * hidden + i] = h2f(table[token_ids[t] * hidden + i]);
}

/* K2 (Skip)RMSNorm: s = x (+ skip); residual_out = fp16(s); y = fp16(fp32(residual) * rsqrt(mean(r^2)+eps) * w) */
REF_API void ref_rmsnorm(const float* x, const float* skip, const f16* w, float eps, int64_t T, int hidden, float* out,
                         float* residual_out) {
#pragma omp parallel
    {
    /* (one scratch row per thread: an alloca inside the loop is only released when the outlined function returns -- 16 KB per
     * token row overflowed the thread stacks from ~4000 rows per step, found by the 6144-token step of config 5, round 4) */
    float* tmp = (float*)malloc(sizeof(float) * hidden);
#pragma omp for
    for (int64_t t = 0; t < T; ++t) {
        const float* xr = x + t * hidden;
        double ss = 0;
        for (int i = 0; i < hidden; ++i) {
            float s = xr[i];
            if (skip) s = rh(s + skip[t * hidden + i]);
            tmp[i] = s;
            ss += (double)s * s;
        }
        if (residual_out) memcpy(residual_out + t * hidden, tmp, sizeof(float) * hidden);
        const float inv = 1.0f / sqrtf((float)(ss / hidden) + eps);
        for (int i = 0; i < hidden; ++i) out[t * hidden + i] = rh(tmp[i] * inv * h2f(w[i]));
    }
    free(tmp);
    }
}

static inline float dot_f32(const float* a, const float* b, int n) {
    if (g_mode & REF_MODE_F64_ACC) {
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        int k = 0;
        for (; k + 4 <= n; k += 4) {
            s0 += (double)a[k] * b[k]; s1 += (double)a[k + 1] * b[k + 1];
            s2 += (double)a[k + 2] * b[k + 2]; s3 += (double)a[k + 3] * b[k + 3];
        }
        for (; k < n; ++k) s0 += (double)a[k] * b[k];
        return (float)((s0 + s1) + (s2 + s3));
    }
    if (g_mode & REF_MODE_ALT_ORDER) { /* two 8-lane accumulators over 16-element steps, lanes summed left to right */
        __m256 c0 = _mm256_setzero_ps(), c1 = _mm256_setzero_ps();
        int k = 0;
        for (; k + 16 <= n; k += 16) {
            c0 = _mm256_fmadd_ps(_mm256_loadu_ps(a + k), _mm256_loadu_ps(b + k), c0);
            c1 = _mm256_fmadd_ps(_mm256_loadu_ps(a + k + 8), _mm256_loadu_ps(b + k + 8), c1);
        }
        float t[8]; _mm256_storeu_ps(t, _mm256_add_ps(c0, c1));
        float s = 0;
        for (int i = 0; i < 8; ++i) s += t[i];
        for (; k < n; ++k) s += a[k] * b[k];
        return s;
    }
    if (g_mode & REF_MODE_ALT_ORDER2) { /* one 8-lane accumulator over 8-element steps, lanes summed pairwise */
        __m256 c0 = _mm256_setzero_ps();
        int k = 0;
        for (; k + 8 <= n; k += 8) c0 = _mm256_fmadd_ps(_mm256_loadu_ps(a + k), _mm256_loadu_ps(b + k), c0);
        float t[8]; _mm256_storeu_ps(t, c0);
        float s = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
        for (; k < n; ++k) s += a[k] * b[k];
        return s;
    }
    __m256 acc0 = _mm256_setzero_ps(), acc1 = _mm256_setzero_ps(), acc2 = _mm256_setzero_ps(), acc3 = _mm256_setzero_ps();
    int k = 0;
    for (; k + 32 <= n; k += 32) {
        acc0 = _mm256_fmadd_ps(_mm256_loadu_ps(a + k), _mm256_loadu_ps(b + k), acc0);
        acc1 = _mm256_fmadd_ps(_mm256_loadu_ps(a + k + 8), _mm256_loadu_ps(b + k + 8), acc1);
        acc2 = _mm256_fmadd_ps(_mm256_loadu_ps(a + k + 16), _mm256_loadu_ps(b + k + 16), acc2);
        acc3 = _mm256_fmadd_ps(_mm256_loadu_ps(a + k + 24), _mm256_loadu_ps(b + k + 24), acc3);
    }
    acc0 = _mm256_add_ps(_mm256_add_ps(acc0, acc1), _mm256_add_ps(acc2, acc3));
    float tmp[8]; _mm256_storeu_ps(tmp, acc0);
    float s = ((tmp[0] + tmp[4]) + (tmp[1] + tmp[5])) + ((tmp[2] + tmp[6]) + (tmp[3] + tmp[7]));
    for (; k < n; ++k) s += a[k] * b[k];
    return s;
}

/* K3/K9/K11 linear: y[m,n] = sum_k x[m,k] * Wdeq[n,k], fp32 accumulate.
 *   W fp16         : Wdeq = fp32(W)
 *   W8A16          : y = scale[n] * sum_k x * int8            (per-output-channel symmetric)
 *   W4A16 group g  : y = sum_G scale[n,G] * sum_{k in G} x * (nibble-8)
 * out_fp32 = 0 rounds the result to fp16 (activations), 1 keeps fp32 (logits, llm_engine.cc:207-222). */
REF_API void ref_quant_act_rows(const float* x, int64_t M, int K, int8_t* q, float* sx);
static void linear_fwd_a8(const ref_linear* l, const float* x, int64_t M, float* y, int out_fp32);

REF_API void ref_linear_fwd(const ref_linear* l, const float* x, int64_t M, float* y, int out_fp32) {
    const int N = l->N, K = l->K;
    if (l->qbit == 8 && l->a8) { linear_fwd_a8(l, x, M, y, out_fp32); return; }
    /* NB weight rows are dequantised at a time and every activation row is multiplied against all of them while it sits in L1: the
     * activations stream through the caches N / NB times instead of N times (round 4: the 6144-token steps of config 5 were
     * memory-bound at ~16 GFLOP/s).  Every dot product is still ONE dot_f32 call: the summation order, hence every bit, is unchanged. */
    enum { NB = 8 };
#pragma omp parallel
    {
        float* wblk = (float*)malloc(sizeof(float) * K * NB);
#pragma omp for schedule(static)
        for (int n0 = 0; n0 < N; n0 += NB) {
            const int nb = N - n0 < NB ? N - n0 : NB;
            for (int j = 0; j < nb; ++j) {
                const int n = n0 + j;
                float* wrow = wblk + (size_t)j * K;
                if (l->qbit == 0) {
                    const f16* w = l->w16 + (size_t)n * K;
                    int k = 0;
                    for (; k + 8 <= K; k += 8) _mm256_storeu_ps(wrow + k, _mm256_cvtph_ps(_mm_loadu_si128((const __m128i*)(w + k))));
                    for (; k < K; ++k) wrow[k] = h2f(w[k]);
                } else if (l->qbit == 8) {
                    const int8_t* w = l->w8 + (size_t)n * K;
                    int k = 0;
                    for (; k + 8 <= K; k += 8)
                        _mm256_storeu_ps(wrow + k, _mm256_cvtepi32_ps(_mm256_cvtepi8_epi32(_mm_loadl_epi64((const __m128i*)(w + k)))));
                    for (; k < K; ++k) wrow[k] = (float)w[k];
                } else {
                    /* W4A16: the dequantised weight IS an fp16 number, w = fp16((nibble - 8) * scale[n, k / group]) (the product is
                     * exact in fp32, so this is one rounding) -- what a W4A16 kernel feeds its fp16 matrix unit; then an fp32 dot */
                    const uint8_t* w = l->w4 + (size_t)n * K / 2;
                    const int G = K / l->group;
                    for (int k = 0; k < K; k += 2) {
                        const float sc = h2f(l->scale[(size_t)n * G + k / l->group]);
                        wrow[k] = rh((float)((int)(w[k / 2] & 15) - 8) * sc);
                        wrow[k + 1] = rh((float)((int)(w[k / 2] >> 4) - 8) * sc);
                    }
                }
            }
            for (int64_t m = 0; m < M; ++m) {
                const float* xr = x + m * K;
                for (int j = 0; j < nb; ++j) {
                    float acc = dot_f32(xr, wblk + (size_t)j * K, K);
                    if (l->qbit == 8) acc *= h2f(l->scale[n0 + j]);
                    y[m * N + n0 + j] = out_fp32 ? acc : rh(acc);
                }
            }
        }
        free(wblk);
    }
}

/* online_i8i8 (W8A8; the mode src/backends/cuda/resource_manager.cc:51-52 hands to ppl.nn, whose kernels are not in
 * the tree -- the arithmetic below is this build's specification, parity with the reference unpinned):
 *   per token row   amax = max|x|, sx = amax / 127 (fp32), q = clamp(rint(x * (127 / amax)), -127, 127)
 *   per weight row  scale = fp16(max|w| / 127), q = clamp(rint(w / scale), -127, 127)      (at load time)
 *   y[m,n] = fp16( (float)(sum_k qx*qw as int32) * sx[m] * scale[n] )                    (multiplied in that order) */
REF_API void ref_quant_act_rows(const float* x, int64_t M, int K, int8_t* q, float* sx) {
#pragma omp parallel for
    for (int64_t m = 0; m < M; ++m) {
        const float* xr = x + m * K;
        float amax = 0.f;
        for (int k = 0; k < K; ++k) { float a = fabsf(xr[k]); if (a > amax) amax = a; }
        const float inv = amax > 0.f ? 127.0f / amax : 0.f;
        sx[m] = amax / 127.0f;
        for (int k = 0; k < K; ++k) {
            float v = rintf(xr[k] * inv);
            v = v > 127.f ? 127.f : (v < -127.f ? -127.f : v);
            q[m * K + k] = (int8_t)v;
        }
    }
}

REF_API void ref_quant_weight_rows(const f16* w, int N, int K, int8_t* q, f16* scale) {
#pragma omp parallel for
    for (int n = 0; n < N; ++n) {
        const f16* wr = w + (size_t)n * K;
        float amax = 0.f;
        for (int k = 0; k < K; ++k) { float a = fabsf(h2f(wr[k])); if (a > amax) amax = a; }
        const f16 sh = f2h(amax / 127.0f);
        scale[n] = sh;
        const float s = h2f(sh) > 0.f ? h2f(sh) : 1.0f;
        for (int k = 0; k < K; ++k) {
            float v = rintf(h2f(wr[k]) / s);
            v = v > 127.f ? 127.f : (v < -127.f ? -127.f : v);
            q[(size_t)n * K + k] = (int8_t)v;
        }
    }
}

static void linear_fwd_a8(const ref_linear* l, const float* x, int64_t M, float* y, int out_fp32) {
    const int N = l->N, K = l->K;
    int8_t* xq = (int8_t*)malloc((size_t)M * K);
    float* sx = (float*)malloc(sizeof(float) * M);
    ref_quant_act_rows(x, M, K, xq, sx);
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n) {
        const int8_t* w = l->w8 + (size_t)n * K;
        const float sw = h2f(l->scale[n]);
        for (int64_t m = 0; m < M; ++m) {
            const int8_t* xr = xq + m * K;
            int32_t acc = 0;
            for (int k = 0; k < K; ++k) acc += (int32_t)xr[k] * (int32_t)w[k];
            const float v = ((float)acc * sx[m]) * sw;
            y[m * N + n] = out_fp32 ? v : rh(v);
        }
    }
    free(xq); free(sx);
}

/* stand-alone int8 x int8 linear for operator tests */
REF_API void ref_linear_i8_raw(const float* x, const int8_t* w, const f16* scale, int64_t M, int N, int K, float* y,
                               int out_fp32) {
    ref_linear l; memset(&l, 0, sizeof(l));
    l.N = N; l.K = K; l.qbit = 8; l.a8 = 1; l.scale = (f16*)scale; l.w8 = (int8_t*)w;
    ref_linear_fwd(&l, x, M, y, out_fp32);
}

/* stand-alone form for operator tests */
REF_API void ref_linear_raw(const float* x, const void* w, const f16* scale, int qbit, int group, int64_t M, int N, int K,
                            float* y, int out_fp32) {
    ref_linear l; memset(&l, 0, sizeof(l));
    l.N = N; l.K = K; l.qbit = qbit; l.group = group; l.scale = (f16*)scale;
    if (qbit == 0) l.w16 = (f16*)w; else if (qbit == 8) l.w8 = (int8_t*)w; else l.w4 = (uint8_t*)w;
    ref_linear_fwd(&l, x, M, y, out_fp32);
}

/* K10 SwiGLU: out = fp16( silu(gate) * up ), gate = gu[:, :inter], up = gu[:, inter:] */
REF_API void ref_silu_mul(const float* gate_up, int64_t T, int inter, float* out) {
#pragma omp parallel for
    for (int64_t t = 0; t < T; ++t)
        for (int i = 0; i < inter; ++i) {
            float g = gate_up[t * 2 * inter + i], u = gate_up[t * 2 * inter + inter + i];
            out[t * inter + i] = rh(g / (1.0f + expf(-g)) * u);
        }
}

/* K4+K5: RoPE (half-split pairing (i, i+D/2), HF convention) on q and k, then write k,v to the cache.
 * Position of row t of request b: start_pos[b] + (t - seq_starts[b])  (llm_generator.cc:263-298).
 * int8 KV, per group of `g` channels: scale = fp16(max|x| / 127); inv = 1 / fp32(scale) (correctly rounded, 0 when
 * scale == 0); q = clamp(rint(x * inv), -127, 127). */
REF_API void ref_rope_kv_write(float* qkv, const float* rope, const ref_model_desc* d, int H, int Hkv, int D, int layer,
                               void* kv_cache, f16* kv_scale, int64_t kv_tokens, const int64_t* seq_starts,
                               const int64_t* start_pos, const int64_t* cache_indices, int64_t max_pages, int64_t B) {
    const int half = D / 2;
    const int64_t row = (int64_t)(H + 2 * Hkv) * D;
    const int g = d->cache_quant_group;
    const kv_strides cs = kv_strides_of(d->cache_layout, kv_tokens, d->num_layers, Hkv, D);
    const kv_strides ss = kv_strides_of(d->cache_layout, kv_tokens, d->num_layers, Hkv, D / (g > 0 ? g : 1));
    for (int64_t b = 0; b < B; ++b) {
#pragma omp parallel for
        for (int64_t t = seq_starts[b]; t < seq_starts[b + 1]; ++t) {
            const int64_t pos = start_pos[b] + (t - seq_starts[b]);
            const float* cs_row = rope + pos * D;
            float* r = qkv + t * row;
            for (int h = 0; h < H + Hkv; ++h) { /* q heads then k heads */
                float* x = r + (int64_t)h * D;
                for (int i = 0; i < half; ++i) {
                    const float c = cs_row[i], s = cs_row[half + i];
                    const float a = x[i], bb = x[i + half];
                    x[i] = rh(a * c - bb * s);
                    x[i + half] = rh(bb * c + a * s);
                }
            }
            const int64_t slot = kv_slot(d, cache_indices, max_pages, b, pos);
            for (int kv = 0; kv < 2; ++kv)
                for (int h = 0; h < Hkv; ++h) {
                    const float* x = r + (int64_t)(H + kv * Hkv + h) * D;
                    const int64_t base = layer * cs.sL + kv * cs.sKV + h * cs.sH + slot * cs.sN;
                    if (d->cache_quant_bit == 0) {
                        if (g_mode & REF_MODE_FP32_ACT) for (int i = 0; i < D; ++i) ((float*)kv_cache)[base + i] = x[i];
                        else for (int i = 0; i < D; ++i) ((f16*)kv_cache)[base + i] = f2h(x[i]);
                    } else {
                        const int64_t sbase = layer * ss.sL + kv * ss.sKV + h * ss.sH + slot * ss.sN;
                        for (int gi = 0; gi < D / g; ++gi) {
                            float mx = 0;
                            for (int i = 0; i < g; ++i) mx = fmaxf(mx, fabsf(x[gi * g + i]));
                            const f16 sh = f2h(mx / 127.0f);
                            const float sf = h2f(sh);
                            kv_scale[sbase + gi] = sh;
                            const float inv = sf > 0 ? 1.0f / sf : 0.0f;
                            for (int i = 0; i < g; ++i) {
                                float qv = rintf(x[gi * g + i] * inv);
                                qv = fminf(fmaxf(qv, -127.0f), 127.0f);
                                ((int8_t*)kv_cache)[base + gi * g + i] = (int8_t)qv;
                            }
                        }
                    }
                }
        }
    }
}

/* K6/K7/K8 MultiHeadCacheAttention: for row t of request b at position pos, head hq (kv head hq / (H/Hkv)):
 *   s_j = (q . K_j) / sqrt(D) for j in [0, pos];  p = softmax(s);  o = fp16( sum_j p_j V_j )
 * K and V are ALWAYS read back from the cache (so prefill, cache-prefill and decode agree by construction). */
REF_API void ref_attention(const float* qkv, const ref_model_desc* d, int H, int Hkv, int D, int layer, const void* kv_cache,
                           const f16* kv_scale, int64_t kv_tokens, const int64_t* seq_starts, const int64_t* start_pos,
                           const int64_t* cache_indices, int64_t max_pages, int64_t B, float* out) {
    const int64_t row = (int64_t)(H + 2 * Hkv) * D;
    const int g = d->cache_quant_group;
    const int grp = H / Hkv;
    const int f32kv = d->cache_quant_bit == 0 && (g_mode & REF_MODE_FP32_ACT);
    const float sm = 1.0f / sqrtf((float)D);
    const kv_strides cs = kv_strides_of(d->cache_layout, kv_tokens, d->num_layers, Hkv, D);
    const kv_strides ss = kv_strides_of(d->cache_layout, kv_tokens, d->num_layers, Hkv, D / (g > 0 ? g : 1));
    for (int64_t b = 0; b < B; ++b) {
        const int64_t nt = seq_starts[b + 1] - seq_starts[b];
#pragma omp parallel for collapse(2) schedule(dynamic)
        for (int64_t ti = 0; ti < nt; ++ti)
            for (int hq = 0; hq < H; ++hq) {
                const int64_t t = seq_starts[b] + ti;
                const int64_t pos = start_pos[b] + ti;
                const int hk = hq / grp;
                const float* q = qkv + t * row + (int64_t)hq * D;
                float* sc = (float*)malloc(sizeof(float) * (pos + 1));
                float* vec = (float*)malloc(sizeof(float) * D);
                double* acc = (double*)calloc(D, sizeof(double));
                float mx = -INFINITY;
                for (int64_t j = 0; j <= pos; ++j) {
                    const int64_t slot = kv_slot(d, cache_indices, max_pages, b, j);
                    const int64_t base = layer * cs.sL + 0 * cs.sKV + hk * cs.sH + slot * cs.sN;
                    if (d->cache_quant_bit == 0) {
                        if (f32kv) for (int i = 0; i < D; ++i) vec[i] = ((const float*)kv_cache)[base + i];
                        else for (int i = 0; i < D; ++i) vec[i] = h2f(((const f16*)kv_cache)[base + i]);
                    }
                    else {
                        const int64_t sbase = layer * ss.sL + 0 * ss.sKV + hk * ss.sH + slot * ss.sN;
                        const int8_t* kq = (const int8_t*)kv_cache + base;   /* (group-blocked: same products, vectorisable) */
                        for (int gi = 0; gi < D / g; ++gi) {
                            const float s1 = h2f(kv_scale[sbase + gi]);
                            for (int i = gi * g; i < (gi + 1) * g; ++i) vec[i] = (float)kq[i] * s1;
                        }
                    }
                    sc[j] = dot_f32(q, vec, D) * sm;
                    mx = fmaxf(mx, sc[j]);
                }
                double den = 0;
                for (int64_t j = 0; j <= pos; ++j) {
                    const float p = expf(sc[j] - mx);
                    den += p;
                    const int64_t slot = kv_slot(d, cache_indices, max_pages, b, j);
                    const int64_t base = layer * cs.sL + 1 * cs.sKV + hk * cs.sH + slot * cs.sN;
                    if (d->cache_quant_bit == 0) {
                        if (f32kv) for (int i = 0; i < D; ++i) acc[i] += (double)p * ((const float*)kv_cache)[base + i];
                        else for (int i = 0; i < D; ++i) acc[i] += (double)p * h2f(((const f16*)kv_cache)[base + i]);
                    }
                    else {
                        const int64_t sbase = layer * ss.sL + 1 * ss.sKV + hk * ss.sH + slot * ss.sN;
                        const int8_t* vq = (const int8_t*)kv_cache + base;
                        const double pd = (double)p;
                        for (int gi = 0; gi < D / g; ++gi) {
                            const float s1 = h2f(kv_scale[sbase + gi]);
                            for (int i = gi * g; i < (gi + 1) * g; ++i) acc[i] += pd * (double)((float)vq[i] * s1);
                        }
                    }
                }
                for (int i = 0; i < D; ++i) out[t * (int64_t)H * D + (int64_t)hq * D + i] = rh((float)(acc[i] / den));
                free(sc); free(vec); free(acc);
            }
    }
}

/* ------------------------------------------------------------------------------------------------
 * the whole forward = Runtime::Run() (src/engine/llm_engine.cc:113-116) for `nranks` tensor-parallel
 * slices simulated in-process: row-parallel outputs are summed in fp32 over ranks and rounded to fp16
 * (the all-reduce), vocab-parallel logits are concatenated (the all-gather).
 * logits_out: fp32 [B, vocab].  hidden_dump (optional): fp16-valued floats [L+1, T, hidden]
 * (residual stream after each layer; index 0 = embeddings) for per-layer fixtures.
 * ---------------------------------------------------------------------------------------------- */
REF_API int ref_forward(ref_model** ranks, int nranks, const ref_step* st, float* logits_out, float* hidden_dump) {
    ref_model* m0 = ranks[0];
    const ref_model_desc* d = &m0->d;
    const int hd = d->hidden_dim;
    const int64_t T = st->num_tokens, B = st->batch;
    const int H = m0->H, Hkv = m0->Hkv, D = m0->D, inter = m0->inter;
    float* h = (float*)malloc(sizeof(float) * T * hd);      /* residual stream */
    float* xn = (float*)malloc(sizeof(float) * T * hd);
    float* part = (float*)malloc(sizeof(float) * T * hd);
    float* sum = (float*)malloc(sizeof(float) * T * hd);
    float* qkv = (float*)malloc(sizeof(float) * T * (H + 2 * Hkv) * D);
    float* att = (float*)malloc(sizeof(float) * T * H * D);
    float* gu = (float*)malloc(sizeof(float) * T * 2 * inter);
    float* act = (float*)malloc(sizeof(float) * T * inter);

    ref_embedding(st->token_inputs, m0->embed, T, hd, h);
    if (hidden_dump) memcpy(hidden_dump, h, sizeof(float) * T * hd);
    float* ffn_sum = (float*)malloc(sizeof(float) * T * hd);
    float* pending = NULL; /* row-parallel FFN output waiting to be folded into the next SkipRMSNorm */
    for (int l = 0; l < d->num_layers; ++l) {
        /* h <- fp16(h + pending); xn = norm(h) */
        ref_rmsnorm(h, pending, m0->layers[l].attn_norm, d->norm_eps, T, hd, xn, h);
        memset(sum, 0, sizeof(float) * T * hd);
        for (int r = 0; r < nranks; ++r) {
            ref_model* m = ranks[r];
            ref_layer* L = &m->layers[l];
            ref_linear_fwd(&L->wqkv, xn, T, qkv, 0);
            ref_rope_kv_write(qkv, m->rope, d, H, Hkv, D, l, m->kv_cache, m->kv_scale, m->kv_tokens, st->seq_starts,
                              st->start_pos, st->cache_indices, st->max_pages, B);
            ref_attention(qkv, d, H, Hkv, D, l, m->kv_cache, m->kv_scale, m->kv_tokens, st->seq_starts, st->start_pos,
                          st->cache_indices, st->max_pages, B, att);
            ref_linear_fwd(&L->wo, att, T, part, 0);
            for (int64_t i = 0; i < T * hd; ++i) sum[i] += part[i];
        }
        if (nranks > 1) for (int64_t i = 0; i < T * hd; ++i) sum[i] = rh(sum[i]); /* all-reduce result is fp16 */
        ref_rmsnorm(h, sum, m0->layers[l].ffn_norm, d->norm_eps, T, hd, xn, h);
        memset(ffn_sum, 0, sizeof(float) * T * hd);
        for (int r = 0; r < nranks; ++r) {
            ref_layer* L = &ranks[r]->layers[l];
            ref_linear_fwd(&L->w13, xn, T, gu, 0);
            ref_silu_mul(gu, T, inter, act);
            ref_linear_fwd(&L->w2, act, T, part, 0);
            for (int64_t i = 0; i < T * hd; ++i) ffn_sum[i] += part[i];
        }
        if (nranks > 1) for (int64_t i = 0; i < T * hd; ++i) ffn_sum[i] = rh(ffn_sum[i]);
        pending = ffn_sum;
        if (hidden_dump)
            for (int64_t i = 0; i < T * hd; ++i) hidden_dump[(size_t)(l + 1) * T * hd + i] = rh(h[i] + ffn_sum[i]);
    }
    /* K11: last-token gather + final (Skip)RMSNorm + lm_head */
    float* hl = (float*)malloc(sizeof(float) * B * hd);
    float* pl = (float*)malloc(sizeof(float) * B * hd);
    for (int64_t b = 0; b < B; ++b) {
        const int64_t t = st->seq_starts[b + 1] - 1;
        memcpy(hl + b * hd, h + t * hd, sizeof(float) * hd);
        if (pending) memcpy(pl + b * hd, pending + t * hd, sizeof(float) * hd);
    }
    float* hn = (float*)malloc(sizeof(float) * B * hd);
    ref_rmsnorm(hl, pending ? pl : NULL, m0->norm, d->norm_eps, B, hd, hn, NULL);
    const int vl = m0->vocab_local;
    float* lg = (float*)malloc(sizeof(float) * B * vl);
    for (int r = 0; r < nranks; ++r) {
        ref_linear_fwd(&ranks[r]->output, hn, B, lg, 1);
        for (int64_t b = 0; b < B; ++b) memcpy(logits_out + b * d->vocab_size + (int64_t)r * vl, lg + b * vl, sizeof(float) * vl);
    }
    free(ffn_sum);
    free(h); free(xn); free(part); free(sum); free(qkv); free(att); free(gu); free(act);
    free(hl); free(pl); free(hn); free(lg);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * sampler: PostProcessor::SampleTopKTopP (src/backends/cuda/post_processor.cc:121-219).  The kernel is
 * external (ppl::kernel::llm::cuda::pmx::sample_topk_topp); DESIGN.md fixes:
 *   x = logits / temperature (temperature NULL or <= 0 -> 1)
 *   top_k == 1: token = first argmax, logprob = x[token] - logsumexp(x)
 *   otherwise : candidates = k largest x (ties: lower index first), p = softmax over candidates,
 *               keep the shortest prefix with cumulative p >= top_p (top_p <= 0 -> keep 1),
 *               renormalise, pick the first candidate whose cumulative p exceeds rand*total;
 *               logprob = x[token] - logsumexp(x) over the full row.
 * ---------------------------------------------------------------------------------------------- */
REF_API void ref_sample(const float* logits, const float* temperatures, const float* top_p, const float* rnd, int batch,
                        int vocab, int stride, int top_k, float default_top_p, int32_t* out_tok, float* out_logprob) {
#pragma omp parallel for
    for (int b = 0; b < batch; ++b) {
        const float* row = logits + (int64_t)b * stride;
        const float temp = (temperatures && temperatures[b] > 0) ? temperatures[b] : 1.0f;
        const float invt = 1.0f / temp;
        float mx = -INFINITY; int am = 0;
        for (int i = 0; i < vocab; ++i) { float x = row[i] * invt; if (x > mx) { mx = x; am = i; } }
        double den = 0;
        for (int i = 0; i < vocab; ++i) den += exp((double)(row[i] * invt - mx));
        const float lse = mx + (float)log(den);
        int tok = am;
        if (top_k != 1) {
            /* top_k <= 0: top-p over the whole vocabulary (probabilities = softmax over ALL logits), candidates capped
             * at 1024; top_k > 1024 is clamped to 1024 */
            const int full = top_k <= 0;
            int k = full ? 1024 : (top_k < 1024 ? top_k : 1024);
            if (k > vocab) k = vocab;
            int* idx = (int*)malloc(sizeof(int) * k);
            float* val = (float*)malloc(sizeof(float) * k);
            int n = 0;
            for (int i = 0; i < vocab; ++i) { /* insertion into a sorted list (desc value, asc index) */
                float x = row[i] * invt;
                if (n < k || x > val[n - 1]) {
                    int p = n < k ? n : k - 1;
                    while (p > 0 && val[p - 1] < x) { val[p] = val[p - 1]; idx[p] = idx[p - 1]; --p; }
                    val[p] = x; idx[p] = i;
                    if (n < k) ++n;
                }
            }
            const float tp = top_p ? top_p[b] : default_top_p;
            double tot = 0;
            if (full) tot = den;
            else for (int i = 0; i < n; ++i) tot += exp((double)(val[i] - mx));
            double cum = 0; int keep = 0;
            for (int i = 0; i < n; ++i) { cum += exp((double)(val[i] - mx)) / tot; keep = i + 1; if (cum >= (double)tp) break; }
            double ktot = 0;
            for (int i = 0; i < keep; ++i) ktot += exp((double)(val[i] - mx));
            const double target = (double)rnd[b] * ktot;
            double c2 = 0; tok = idx[keep - 1];
            for (int i = 0; i < keep; ++i) { c2 += exp((double)(val[i] - mx)); if (c2 > target) { tok = idx[i]; break; } }
            free(idx); free(val);
        }
        out_tok[b] = tok;
        out_logprob[b] = row[tok] * invt - lse;
    }
}

/* penalty: PostProcessor::ApplyPenalty (post_processor.cc:221-281; kernel external).  DESIGN.md fixes:
 * the uint16 count map row batch_slots[b] counts every token the request has fed the model; when
 * start_pos[b] == 0 the row is cleared first; then logits[v] of counted tokens:
 *   x = x > 0 ? x / rep : x * rep ;  x -= presence ;  x -= frequency * count ;  finally all x /= temperature. */
REF_API void ref_penalty(float* logits, const float* temperatures, const float* rep, const float* presence,
                         const float* frequency, const int64_t* batch_slots, const int64_t* token_inputs,
                         const int64_t* seq_starts, const int64_t* start_pos, int batch, int vocab, uint16_t* count_map) {
    for (int b = 0; b < batch; ++b) {
        uint16_t* cm = count_map + batch_slots[b] * (int64_t)vocab;
        if (start_pos[b] == 0) memset(cm, 0, sizeof(uint16_t) * vocab);
        for (int64_t t = seq_starts[b]; t < seq_starts[b + 1]; ++t)
            if (cm[token_inputs[t]] < 65535) cm[token_inputs[t]]++;
        float* row = logits + (int64_t)b * vocab;
        const float temp = (temperatures && temperatures[b] > 0) ? temperatures[b] : 1.0f;
        for (int v = 0; v < vocab; ++v) {
            float x = row[v];
            if (cm[v]) {
                const float r = rep ? rep[b] : 1.0f;
                x = x > 0 ? x / r : x * r;
                if (presence) x -= presence[b];
                if (frequency) x -= frequency[b] * (float)cm[v];
            }
            row[v] = x / temp;
        }
    }
}

REF_API int ref_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
