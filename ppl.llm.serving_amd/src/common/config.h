// Configuration of the hot path.  The three aggregates and their member names are the reference's (src/common/config.h:27-85),
// because the tools fill them straight from the command line (tools/offline_inference.cc:92-134) and the generator / engine
// read them by name; the members are documented and grouped here by who consumes them.  params.json -> ModelConfig is
// src/common/config.cc:31-148; the extra ModelConfig members describe what the reference keeps inside the exported graph.
#pragma once
#include <stdint.h>

#include <set>
#include <string>

namespace ppl { namespace llm {

// What the backend's resource manager needs to bring a model up (HipResourceManager::Init).
struct ResourceConfig final {
    // ---- where the model is --------------------------------------------------------------------------------------
    std::string model_type;        // --model-type        ("llama")
    std::string model_format;      // --model-format      ("pplhip": model_slice_<rank>/weights.pplhip)
    std::string model_dir;         // --model-dir
    std::string model_param_path;  // --model-param-path  (params.json)
    // ---- how it is laid out on the devices -----------------------------------------------------------------------
    int32_t tensor_parallel_size = 0;     // --tensor-parallel-size, a power of two (offline_inference.cc:136-139)
    float max_tokens_scale = 0.f;         // --max-tokens-scale: share of the free memory given to the KV slab
    int32_t max_running_batch = 0;        // --max-running-batch: sizes the sampler and the per-request device arrays
    int32_t max_tokens_per_step = 8192;   // --max-tokens-per-step: sizes the activation buffers (hip backend)
    bool enable_penalty = false;          // --enable-penalty: allocates the uint16 count map [max_running_batch, vocab]
    // ---- hip backend only ------------------------------------------------------------------------------------------
    bool synthetic_weights = false;              // --synthetic-weights: device-side generator instead of loading slices
    uint64_t synthetic_seed = 1234;              // --synthetic-seed
    int64_t synthetic_decisive_head = 0;         // --synthetic-decisive-head N: lm_head row v = embedding row v - N (pplhip_rank_tie_output); 0: off
    uint64_t kv_cache_max_tokens_override = 0;   // --kv-cache-max-tokens: > 0 pins the slab size (tests, benchmarks)
    // ---- engine options of the reference's command line; the hip backend honours the last three -------------------
    struct EngineConfig {
        std::string cublas_layout_hint = "default";   // accepted, ignored
        bool disable_graph_fusion = false;            // accepted, ignored (fusion is static here)
        bool disable_decoding_shm_mha = false;        // accepted, ignored
        bool disable_decoding_inf_mha = false;        // accepted, ignored
        bool disable_decoding_inf_gqa = false;        // accepted, ignored
        int32_t configure_decoding_attn_split_k = 1;  // 0 off / 1 heuristic / 2 always
        int32_t specify_decoding_attn_tpb = 0;        // 0 heuristic / 256 / 512
        std::string quant_method = "none";            // or "online_i8i8" (W8A8)
    };
    EngineConfig engine_config;
};

// What LLMGenerator needs (src/generator/llm_generator.cc:114-191 checks and uses them).
struct GeneratorConfig final {
    // ---- sampling defaults handed to the PostProcessor ---------------------------------------------------------------
    float top_p = 0.0f;                          // --top-p
    int32_t top_k = 1;                           // --top-k
    bool enable_penalty = false;                 // --enable-penalty
    // ---- admission limits ----------------------------------------------------------------------------------------
    int32_t max_running_batch = 0;               // --max-running-batch
    int32_t max_input_tokens_per_request = 0;    // --max-input-tokens-per-request  (longer prompts are rejected)
    int32_t max_output_tokens_per_request = 0;   // --max-output-tokens-per-request (generation length is clamped)
    int32_t max_total_tokens_per_request = 0;    // --max-total-tokens-per-request  (prompt + generation clamp)
    int32_t max_tokens_per_step = 0;             // --max-tokens-per-step: token budget of one step
    int32_t max_prefill_batch = 0;               // --max-prefill-batch (1 with the prefix cache, offline_inference.cc:97-99)
    int max_cooldown_request = 2;                // --max-cooldown-request: finished requests to wait for when KV is full
    bool enable_prefix_cache = false;            // --enable-prefix-cache (needs cache_mode 1)
    // ---- token classes -------------------------------------------------------------------------------------------
    std::set<int> stop_tokens;                   // --stop-tokens: EOS-like tokens
    std::set<int> special_tokens;                // --special_tokens: reported as Response::is_special
    // ---- diagnostics -----------------------------------------------------------------------------------------------
    bool enable_profiling = false;               // --enable-profiling: Connection::OnProfiling once per second
};

// The model as params.json describes it.
struct ModelConfig final {
    // ---- transformer dimensions (required keys) ------------------------------------------------------------------
    int32_t hidden_dim = 0;
    int32_t intermediate_dim = 0;
    int32_t num_layers = 0;
    int32_t num_heads = 0;
    int32_t num_kv_heads = 0;          // optional key, defaults to num_heads
    int32_t vocab_size = 0;
    // ---- KV cache format (required keys; llm_engine.cc:118-169 gives the four layouts) ----------------------------
    int32_t cache_quant_bit = 0;       // 0 (fp16) or 8 (int8)
    int32_t cache_quant_group = 0;     // 8 with int8, 1 with fp16
    int32_t cache_layout = 0;          // 0..3
    int32_t cache_mode = 0;            // 0 contiguous ranges, 1 pages
    int32_t page_size = 0;             // required when cache_mode == 1
    bool dynamic_batching = true;      // required key, must be true
    bool auto_causal = true;           // required key
    // ---- optional keys of this build (inside the exported graph in the reference) ----------------------------------
    float norm_eps = 1e-5f;
    float rope_theta = 10000.f;
    int32_t max_position = 8192;
    int32_t weight_quant_bit = 0;      // 0 / 8 (W8A16) / 4 (W4A16)
    int32_t weight_quant_group = 128;  // W4A16 group size along K
};

bool ParseModelConfig(const std::string& model_param_path, ModelConfig* model_config);
bool ParseModelConfigFromString(const std::string& json_text, ModelConfig* model_config);

}}  // namespace ppl::llm
