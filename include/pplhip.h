/*
 * pplhip.h -- C ABI of the MI355X (gfx950) backend for ppl.llm.serving's batched decode hot path.
 *
 * This is the drop-in boundary.  Everything below the dashed line of the
 * reference's layer map (ppl.nn Runtime/Tensor/Engine, ppl.llm.kernel.cuda, NCCL) is replaced by this
 * library; everything above it (src/engine, src/generator, tools) calls it through `src/backends/hip`.
 * Plain C structs of ints / pointers; no C++ or torch types cross this boundary.
 *
 * Conventions
 *   - every entry point returns 0 on success or a negative pplhip_status; the text of the last error of
 *     a rank is available from pplhip_last_error().  The codes map 1:1 onto the ppl::common::RetCode
 *     values the reference uses in-tree (src/backends/cuda/resource_manager.cc, post_processor.cc).
 *   - the caller owns all host buffers; the library owns all device memory.
 *   - `rank` is the LOCAL rank index inside this context.  A rank's entry points are called from that
 *     rank's worker thread only -- exactly how utils::ParallelExecute drives ranks in the reference
 *     (src/utils/utils.h:37-52).  pplhip_sample/pplhip_penalty act on local rank 0's stream and may be
 *     called from the generator thread (src/engine/llm_engine.cc:204-224).
 *   - no global state: several contexts may coexist.
 *   - set_inputs/run are asynchronous (stream-ordered); pplhip_sample is the one synchronisation point
 *     of a step (src/backends/cuda/post_processor.cc:196).
 */
#ifndef PPLHIP_H_
#define PPLHIP_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPLHIP_API __attribute__((visibility("default")))

/* ---- status codes (ppl::common::RetCode counterparts) ------------------------------------------- */
typedef enum pplhip_status {
    PPLHIP_SUCCESS = 0,
    PPLHIP_OTHER_ERROR = -1,          /* RC_OTHER_ERROR */
    PPLHIP_INVALID_VALUE = -2,        /* RC_INVALID_VALUE */
    PPLHIP_OUT_OF_MEMORY = -3,        /* RC_OUT_OF_MEMORY */
    PPLHIP_DEVICE_RUNTIME_ERROR = -4, /* RC_DEVICE_RUNTIME_ERROR */
    PPLHIP_DEVICE_MEMORY_ERROR = -5,  /* RC_DEVICE_MEMORY_ERROR */
    PPLHIP_NOT_FOUND = -6,            /* RC_NOT_FOUND */
    PPLHIP_UNSUPPORTED = -7           /* RC_UNSUPPORTED */
} pplhip_status;

/* ---- model description: params.json (src/common/config.cc:31-148) + what the exported graph encodes */
typedef struct pplhip_model_desc {
    int32_t hidden_dim;
    int32_t intermediate_dim;
    int32_t num_layers;
    int32_t num_heads;
    int32_t num_kv_heads;
    int32_t vocab_size;
    float norm_eps;          /* RMSNorm epsilon (graph attribute in the reference; 1e-5 for LLaMA-2) */
    float rope_theta;        /* 10000 for LLaMA-2 */
    int32_t max_position;    /* size of the host-built cos/sin table */
    int32_t cache_quant_bit;   /* 0 (fp16 KV) or 8 (int8 KV)            src/generator/llm_generator.cc:131-136 */
    int32_t cache_quant_group; /* 1 or 8 */
    int32_t cache_layout;      /* 0..3                                  src/engine/llm_engine.cc:122-166 */
    int32_t cache_mode;        /* 0 contiguous ranges, 1 paged          src/generator/llm_generator.cc:486-560 */
    int32_t page_size;         /* tokens per page when cache_mode == 1 */
    int32_t weight_quant_bit;   /* 0 fp16 weights, 8 = W8A16 per-output-channel, 4 = W4A16 grouped */
    int32_t weight_quant_group; /* K-group of W4A16 (128); ignored otherwise */
    int32_t act_quant_bit;      /* 0 = fp16 activations; 8 (with weight_quant_bit 8) = online_i8i8, the W8A8 mode of
                                   src/backends/cuda/resource_manager.cc:51-52: per-token int8 activations in front of
                                   every layer linear, int8 x int8 -> int32 on the matrix cores.  In this mode an fp16
                                   [N,K] matrix handed to pplhip_rank_set_tensor for an int8 linear is quantised per
                                   output row on the device ("online"). */
} pplhip_model_desc;

/* ---- context options: what CudaResourceManager::Init/InitTask take (resource_manager.cc:213-428) */
typedef struct pplhip_opts {
    int32_t n_local_ranks;     /* ranks (GPUs) driven by THIS process: tensor_parallel_size in the
                                  reference's single-process mode, 1 in one-process-per-GPU mode */
    int32_t world_size;        /* total tensor-parallel degree across processes */
    int32_t rank_base;         /* global rank of local rank 0 */
    const int32_t* device_ids; /* n_local_ranks HIP device ordinals; NULL = 0..n-1 */
    const void* nccl_unique_id;/* PPLHIP_UNIQUE_ID_BYTES bytes, shared by all processes; NULL when
                                  world_size == n_local_ranks (single process: ncclCommInitAll) */
    int32_t max_running_batch; /* --max-running-batch */
    int32_t max_tokens_per_step; /* --max-tokens-per-step */
    int32_t enable_penalty;    /* --enable-penalty */
    int32_t decoding_attn_split_k; /* --configure-decoding-attn-split-k: 0 off, 1 heuristic, 2 always */
    int32_t decoding_attn_tpb;     /* --specify-decoding-attn-tpb: 0 heuristic, 256, 512 */
    int32_t enable_profiling;  /* record HIP events around the dominant kernels (see pplhip_profile_*): 1 = every class,
                                  2 = decode attention and the whole run only (an event record is a barrier packet) */
} pplhip_opts;

#define PPLHIP_UNIQUE_ID_BYTES 128

/* ---- one step's inputs: ModelInput (src/engine/llm_engine.h:40-60) as the 11-input contract
 *      of the runtime (src/engine/llm_engine.h:124-138, src/engine/llm_engine.cc:29-111) ------------ */
typedef struct pplhip_step {
    int64_t batch;               /* B = start_pos.size() */
    int64_t num_tokens;          /* T = token_inputs.size() */
    int64_t decoding_batches;    /* first rows with seqlen 1 */
    int64_t max_seq_len;
    int64_t max_kv_len;
    int64_t max_pages;           /* cache_mode 1 only */
    const int64_t* token_inputs; /* [T] */
    const int64_t* seq_starts;   /* [B+1] */
    const int64_t* kv_starts;    /* [B+1] */
    const int64_t* start_pos;    /* [B] */
    const int64_t* cache_indices;/* mode 0: [B] first KV slot of each request;
                                    mode 1: [B, max_pages] page ids (INT64_MAX padded), may be NULL when
                                    req_list_changed == 0 (src/engine/llm_engine.cc:67-71) */
    int32_t req_list_changed;
} pplhip_step;

/* ---- sampler arguments: PostProcessor::SampleTopKTopP (src/common/post_processor.h:31-35) ------- */
typedef struct pplhip_sample_args {
    const float* temperatures;   /* host [B] or NULL */
    const int32_t* top_k;        /* host [B] or NULL (unused by the reference kernel, Q3 in SURVEY.md) */
    const float* top_p;          /* host [B] or NULL */
    int32_t batch;
    int32_t vocab_size;
    int32_t batch_stride;        /* logits row stride in floats */
    int32_t default_top_k;
    float default_top_p;
    int32_t req_list_changed;
    int32_t enable_penalty;
} pplhip_sample_args;

/* ---- penalty arguments: PostProcessor::ApplyPenalty (src/common/post_processor.h:37-42) --------- */
typedef struct pplhip_penalty_args {
    const float* temperatures;          /* host [B] */
    const float* repetition_penalties;  /* host [B] */
    const float* presence_penalties;    /* host [B] or NULL */
    const float* frequency_penalties;   /* host [B] or NULL */
    const int64_t* batch_slots;         /* host [B] */
    int32_t batch;
    int32_t vocab_size;
    int32_t req_list_changed;
} pplhip_penalty_args;

typedef struct pplhip_ctx pplhip_ctx;

/* ================================================================================================
 * life cycle  -- replaces CudaResourceManager::Init / ~CudaResourceManager
 *               (src/backends/cuda/resource_manager.cc:373-428, resource_manager.h:86-109)
 * ============================================================================================== */

/* library/ABI version (major<<16 | minor) */
PPLHIP_API int pplhip_version(void);

/* number of visible HIP devices, or a negative status (replaces the cudaGetDeviceCount probe). */
PPLHIP_API int pplhip_device_count(void);

/* fills PPLHIP_UNIQUE_ID_BYTES bytes with an RCCL unique id (replaces ppl::common::InitNccl,
 * resource_manager.cc:393, for the one-process-per-GPU launch). */
PPLHIP_API int pplhip_get_unique_id(void* out);

/* creates streams, RCCL communicators, cos/sin table and per-rank scratch.  Device work of rank r
 * happens on opts->device_ids[r]. */
PPLHIP_API int pplhip_init(const pplhip_model_desc* desc, const pplhip_opts* opts, pplhip_ctx** out);

PPLHIP_API void pplhip_destroy(pplhip_ctx* ctx);

/* ---- tensor-parallel collectives (replaces the NCCL calls ppl.nn issues on the engine's stream,
 *      src/backends/cuda/resource_manager.cc:239,392-398) ------------------------------------------------
 * Two implementations behind the same step schedule: RCCL, and direct kernels over peer-mapped memory that use all
 * xGMI links of a rank at once (csrc/k_comm.hip).  The direct path is preferred whenever its start-up self-test passes
 * on EVERY rank (env PPLHIP_COMM=auto|p2p|rccl).  When all ranks live in one process (the reference's mode)
 * pplhip_init connects them itself.  With one process per GPU the launcher does, after pplhip_init on every process:
 *     pplhip_comm_export(ctx, 0, mine);  all-gather the handles by global rank;  pplhip_comm_connect(ctx, all);
 * pplhip_comm_connect is collective (every process calls it at about the same time); skipping it keeps RCCL. */
#define PPLHIP_IPC_HANDLE_BYTES 64
PPLHIP_API int pplhip_comm_export(pplhip_ctx* ctx, int rank, void* handle_out /* PPLHIP_IPC_HANDLE_BYTES */);
PPLHIP_API int pplhip_comm_connect(pplhip_ctx* ctx, const void* all_handles /* world_size x PPLHIP_IPC_HANDLE_BYTES */);
/* collectives in use: 0 none (single rank), 1 RCCL, 2 direct kernels over peer-mapped memory */
PPLHIP_API int pplhip_comm_mode(pplhip_ctx* ctx);
/* 1 when the all-reduces behind wo / w2 also do the residual add and the RMSNorm that consumes them, on the rows a rank owns between the two
 * shots of the direct all-reduce (sequence-parallel residual stream: 1 / world_size of the norm work per rank, one launch less per
 * half-layer; the same bytes on the links).  Direct collectives only; RCCL keeps all-reduce + replicated norm.  PPLHIP_TP_FUSE_NORM=0: off */
PPLHIP_API int pplhip_comm_fused_norm(pplhip_ctx* ctx);

/* What a multi-GPU run actually does, for the benchmark's report and for diagnosing a first run on real links (no counterpart in the
 * reference, whose NCCL set-up either works or aborts: src/backends/cuda/resource_manager.cc:392-422).  Fallbacks are never silent: each
 * one is printed on stderr when it is taken and kept in `notes` (direct collectives -> RCCL when the self-test fails on any rank;
 * two-stream decode -> one stream when the second RCCL communicator cannot be created). */
typedef struct pplhip_comm_info_t {
    int32_t mode;           /* pplhip_comm_mode */
    int32_t selftest;       /* direct collectives' start-up self-test: 0 not run, 1 passed on every rank, -1 failed */
    int32_t schedule;       /* of a pure-decode step of `rows` rows: 0 collectives in-stream, 1 two half-batches on two streams, 2 two chunks
                               with the collectives on the communication stream */
    int32_t has_rccl;       /* an RCCL communicator exists */
    int64_t dual_min_rows, dual_max_rows;   /* row window of the two-stream schedule (0, 0: off) */
    char notes[512];
} pplhip_comm_info_t;
PPLHIP_API int pplhip_comm_info(pplhip_ctx* ctx, int64_t rows, pplhip_comm_info_t* out);
/* average microseconds of one all-reduce of fp16 [rows, hidden_dim] (the step's own message) on `rank`'s stream over `iters` calls;
 * path 0: the collectives in use, 1: RCCL.  *us = -1 when that path does not exist.  Collective: every rank calls it alike -- the call
 * enqueues and synchronises, so a context holding several local ranks calls it from one thread per rank at the same time. */
PPLHIP_API int pplhip_comm_allreduce_us(pplhip_ctx* ctx, int rank, int64_t rows, int32_t iters, int32_t path, float* us);

PPLHIP_API const char* pplhip_last_error(pplhip_ctx* ctx, int rank);

/* ================================================================================================
 * weights -- replaces onnx::RuntimeBuilder::LoadModel of model_slice_<rank>/model.onnx
 *            (src/backends/cuda/resource_manager.cc:117-147,280-289)
 * ============================================================================================== */

/* loads `<slice_dir>/weights.pplhip` (format: DESIGN.md "weight container"). */
PPLHIP_API int pplhip_rank_load(pplhip_ctx* ctx, int rank, const char* slice_dir);

/* uploads one named tensor of this rank's slice from host memory (names: DESIGN.md). */
PPLHIP_API int pplhip_rank_set_tensor(pplhip_ctx* ctx, int rank, const char* name, const void* data,
                                      uint64_t bytes);

/* fills every weight of this rank's slice on the device from the counter-based generator that
 * oracle/llama_ref.c restates (synthetic weights for benchmarks and parity tests). */
PPLHIP_API int pplhip_rank_init_synthetic(pplhip_ctx* ctx, int rank, uint64_t seed);

/* ================================================================================================
 * KV cache slab -- replaces the cudaMemGetInfo/cudaMalloc block of InitTask
 *                  (src/backends/cuda/resource_manager.cc:329-362)
 * ============================================================================================== */

/* bytes one token occupies in the KV slab / the scale slab of one rank (resource_manager.cc:381-387). */
PPLHIP_API int pplhip_kv_block_bytes(pplhip_ctx* ctx, uint64_t* cache_bytes, uint64_t* scale_bytes);

/* kv_cache_max_tokens = max_tokens_scale * free_bytes(local rank 0) * kb/(kb+sb) / kb. */
PPLHIP_API int pplhip_kv_capacity(pplhip_ctx* ctx, float max_tokens_scale, uint64_t* tokens);

/* allocates the slab (+ scale slab) for `tokens` tokens on this rank. */
PPLHIP_API int pplhip_kv_alloc(pplhip_ctx* ctx, int rank, uint64_t tokens);

/* device pointers of the slabs (Resource::items[rank].kv_cache_mem / kv_scale_mem). */
PPLHIP_API int pplhip_kv_ptrs(pplhip_ctx* ctx, int rank, void** kv_cache_mem, void** kv_scale_mem);

/* test/debug: copies `bytes` from the slab (which = 0) or the scale slab (which = 1), starting at byte
 * `offset`, to/from host memory.  Synchronous. */
PPLHIP_API int pplhip_kv_read(pplhip_ctx* ctx, int rank, int which, uint64_t offset, void* dst, uint64_t bytes);
PPLHIP_API int pplhip_kv_write(pplhip_ctx* ctx, int rank, int which, uint64_t offset, const void* src,
                               uint64_t bytes);

/* test/benchmark: fills the whole slab with pseudo-random history (int8 bytes / fp16 values from the
 * counter-based generator, scales in [0.01, 0.03)) so that decode steps can be measured at a given context
 * length without running the prefill first. */
PPLHIP_API int pplhip_kv_fill_synthetic(pplhip_ctx* ctx, int rank, uint64_t seed);

/* ================================================================================================
 * the step -- replaces SetInputTask / RunModelTask (src/engine/llm_engine.cc:29-116)
 * ============================================================================================== */

/* stages the step's small integer arrays into pinned memory and issues ONE async H2D copy. */
PPLHIP_API int pplhip_set_inputs(pplhip_ctx* ctx, int rank, const pplhip_step* step);

/* the decoder forward for the packed ragged batch: Runtime::Run() (llm_engine.cc:115).
 * cache_prefill = ENGINE_CONF_CACHE_PREFILL (llm_engine.cc:114). */
PPLHIP_API int pplhip_run(pplhip_ctx* ctx, int rank, int cache_prefill);

/* device pointer + row stride (floats) of `logits fp32[B, vocab]` of the last run (llm_engine.cc:207-222). */
PPLHIP_API int pplhip_logits(pplhip_ctx* ctx, int rank, float** logits_device, int64_t* stride);

/* test/debug: synchronises the rank's stream and copies the logits [batch, vocab] to host. */
PPLHIP_API int pplhip_copy_logits(pplhip_ctx* ctx, int rank, float* dst, int64_t batch);

/* blocks until the rank's stream is idle. */
PPLHIP_API int pplhip_sync(pplhip_ctx* ctx, int rank);

/* ================================================================================================
 * sampler -- replaces CudaPostProcessor (src/backends/cuda/post_processor.cc:71-281)
 * ============================================================================================== */

PPLHIP_API int pplhip_sample(pplhip_ctx* ctx, const float* logits_device, const pplhip_sample_args* args,
                             int32_t* output_host, float* logprob_host);

/* in-place penalty on the logits of the last run, using the step's device-resident token_inputs /
 * seq_starts / start_pos (llm_engine.cc:204-216). */
PPLHIP_API int pplhip_penalty(pplhip_ctx* ctx, float* logits_device, const pplhip_penalty_args* args);

/* ================================================================================================
 * measurement
 * ============================================================================================== */

/* kernel classes that are timed with HIP events when opts.enable_profiling != 0 */
enum { PPLHIP_PROF_ATTN_DECODE = 0, PPLHIP_PROF_ATTN_PREFILL = 1, PPLHIP_PROF_GEMM = 2,
       PPLHIP_PROF_RUN = 3, PPLHIP_PROF_COUNT = 4 };

/* clears the event log of this rank. */
PPLHIP_API int pplhip_profile_reset(pplhip_ctx* ctx, int rank);

/* synchronises, then returns number of launches and summed duration (ms) of a kernel class since reset. */
PPLHIP_API int pplhip_profile_get(pplhip_ctx* ctx, int rank, int kernel_class, int64_t* launches, double* total_ms);

/* changes opts.enable_profiling (0, 1, 2) for the steps that follow: bench.py times its steps in mode 2 (decode attention
 * through its own dispatch packet, no barrier packets on the stream) and then brackets every kernel class on a few extra,
 * untimed steps to report the GEMM share. */
PPLHIP_API int pplhip_profile_mode(pplhip_ctx* ctx, int mode);

/* free / total device memory of the rank's device (cudaMemGetInfo in llm_generator.cc:777). */
PPLHIP_API int pplhip_mem_info(pplhip_ctx* ctx, int rank, uint64_t* free_bytes, uint64_t* total_bytes);

/* ================================================================================================
 * TEST AND HARNESS SUPPORT -- not part of the hot path, called by no backend's step (the product boundary is the
 * sections above).  Everything from here to the end of the header exists for the parity tests, the diagnosis tools and the
 * benchmark harnesses: the two entry points below and the single-operator entry points that follow.
 * ============================================================================================== */

/* a synthetic model whose greedy answers have a wide top-2 margin (token t is followed by t + shift): the embedding table regenerated
 * from `seed` at amplitude embed_amp (0: kept) and output.weight[v] := tok_embeddings.weight[(v - shift) mod vocab] on this rank's shard.
 * For harness checks that compare the answers of two runs token for token (tools/benchmark_prefix_cache_offline
 * --synthetic-decisive-head; the reference compares nothing: benchmark_prefix_cache_offline.cc:442-508). */
PPLHIP_API int pplhip_rank_tie_output(pplhip_ctx* ctx, int rank, int64_t shift, uint64_t seed, float embed_amp);

/* Diagnosis (tests bisect a logits difference per layer with it): pplhip_run with the residual stream captured after every
 * layer in the oracle's convention: out[0] = embeddings, out[l+1] = fp16(h + FFN output of layer l), fp32 [L+1, T, hidden].
 * Launches eagerly and synchronises the rank's stream.  Not used by any backend. */
PPLHIP_API int pplhip_debug_run_dump(pplhip_ctx* ctx, int rank, float* hidden_dump_host);

/* ================================================================================================
 * single-operator entry points (device pointers, caller-provided stream; used by the parity tests to
 * check each hand-written kernel against the oracle in isolation).  `stream` is a hipStream_t or NULL.
 * All fp16 buffers are IEEE binary16.  Semantics: DESIGN.md "numerics".
 * ============================================================================================== */

PPLHIP_API int pplhip_op_embedding(void* stream, const int64_t* token_ids, const void* table, int64_t T,
                                   int32_t hidden, void* out);

/* out = rmsnorm(x (+ skip)) * w ; if skip != NULL also writes residual_out = fp16(x + skip). */
PPLHIP_API int pplhip_op_rmsnorm(void* stream, const void* x, const void* skip, const void* w, float eps,
                                 int64_t T, int32_t hidden, void* out, void* residual_out);

/* y[M,N] = x[M,K] . W[N,K]^T ; wq_bit 0: W fp16; 8: W int8 + scale fp16[N]; 4: W int4 packed + scale
 * fp16[N, K/group].  out_fp32 != 0 writes float, else fp16. */
PPLHIP_API int pplhip_op_linear(void* stream, const void* x, const void* w, const void* scale, int32_t wq_bit,
                                int32_t group, int64_t M, int32_t N, int32_t K, void* y, int32_t out_fp32);

/* fused K3 + K10: W rows interleaved (gate_0, up_0, gate_1, up_1, ...), N = 2 * inter; y[M, N/2] = silu(gate) * up. */
PPLHIP_API int pplhip_op_linear_swiglu(void* stream, const void* x, const void* w, const void* scale, int32_t wq_bit,
                                       int32_t group, int64_t M, int32_t N, int32_t K, void* y);

/* online_i8i8 (W8A8, src/backends/cuda/resource_manager.cc:51-52).  Per-token activation quantisation: q[M,K] int8,
 * sx[M] = max|x| / 127; per-output-row weight quantisation of an fp16 [N,K] matrix: q[N,K] int8, scale[N] fp16;
 * y[m,n] = (sum_k xq * w as int32) * sx[m] * scale[n], rounded to fp16 (or kept fp32); swiglu as in pplhip_op_linear_swiglu. */
/* (Skip)RMSNorm whose output goes straight to the int8 operand of the next linear (what the runtime launches in front of wqkv / w13
 * in online_i8i8 mode): q[T,hidden] / sx[T] = pplhip_op_quant_act(pplhip_op_rmsnorm(...)), bit for bit */
PPLHIP_API int pplhip_op_rmsnorm_quant(void* stream, const void* x, const void* skip, const void* w, float eps, int64_t T,
                                       int32_t hidden, void* residual_out, void* q, float* sx);
PPLHIP_API int pplhip_op_quant_act(void* stream, const void* x, int64_t M, int32_t K, void* q, float* sx);
PPLHIP_API int pplhip_op_quant_weight(void* stream, const void* w, int32_t N, int32_t K, void* q, void* scale);
PPLHIP_API int pplhip_op_linear_i8(void* stream, const void* xq, const float* sx, const void* w, const void* scale, int64_t M,
                                   int32_t N, int32_t K, void* y, int32_t out_fp32, int32_t swiglu);

PPLHIP_API int pplhip_op_silu_mul(void* stream, const void* gate_up, int64_t T, int32_t inter, void* out);

/* description of a KV slab for the attention / cache-write operators */
typedef struct pplhip_kv_view {
    void* cache;          /* fp16 or int8 */
    void* scale;          /* fp16, NULL when quant_bit == 0 */
    int64_t max_tokens;   /* N */
    int32_t num_layers;   /* L */
    int32_t kv_heads;     /* h (per rank) */
    int32_t head_dim;     /* d */
    int32_t quant_bit, quant_group, layout, mode, page_size;
    int32_t layer;        /* layer this call addresses */
} pplhip_kv_view;

/* RoPE on q,k of the fused qkv[T, (H+2Hkv)*D] (in place on q) + write of k,v into the cache. */
PPLHIP_API int pplhip_op_rope_kv_write(void* stream, void* qkv, const float* cos_sin, const pplhip_kv_view* kv,
                                       const int64_t* seq_starts, const int64_t* start_pos,
                                       const int64_t* cache_indices, int64_t max_pages, int64_t B, int64_t T,
                                       int32_t num_heads);

/* attention over the cache for rows [row_begin, row_end) of the batch: decode rows (seqlen 1) go to
 * the decode kernel, others to the prefill kernel.  out[T, H*D] fp16. */
PPLHIP_API int pplhip_op_attention(void* stream, const void* qkv, const pplhip_kv_view* kv,
                                   const int64_t* seq_starts, const int64_t* start_pos,
                                   const int64_t* cache_indices, int64_t max_pages, int64_t B, int64_t T,
                                   int64_t decoding_batches, int64_t max_seq_len, int64_t max_kv_len,
                                   int32_t num_heads, int32_t split_k, void* workspace, uint64_t workspace_bytes,
                                   void* out);

/* builds the fp32 cos/sin table [max_position, head_dim] (cos first half, sin second half per row). */
PPLHIP_API int pplhip_build_rope_table(float* host_out, int32_t max_position, int32_t head_dim, float theta);

#ifdef __cplusplus
}
#endif

#endif /* PPLHIP_H_ */
