// Configuration of the hot path.  Field names and meaning follow the reference so that its tools' flag handling maps
// one to one: ResourceConfig / GeneratorConfig / ModelConfig of src/common/config.h:27-85; params.json keys of
// src/common/config.cc:31-148.  Extra ModelConfig fields describe what the reference keeps inside the exported graph.
#pragma once
#include <stdint.h>

#include <set>
#include <string>

namespace ppl { namespace llm {

struct ResourceConfig final {
    std::string model_type;
    std::string model_format;
    std::string model_dir;
    std::string model_param_path;
    int32_t tensor_parallel_size = 0;
    float max_tokens_scale = 0.f;
    int32_t max_running_batch = 0;
    int32_t max_tokens_per_step = 8192;  // sizes the backend's activation buffers (hip backend only)
    bool enable_penalty = false;
    bool synthetic_weights = false;      // hip backend: fill the slices with the synthetic generator instead of loading
    uint64_t synthetic_seed = 1234;
    uint64_t kv_cache_max_tokens_override = 0;  // hip backend: > 0 pins the slab size (tests, benchmarks)
    struct EngineConfig {
        std::string cublas_layout_hint = "default";  // accepted for CLI compatibility, ignored by the hip backend
        bool disable_graph_fusion = false;           // idem (fusion is static in the hip backend)
        bool disable_decoding_shm_mha = false;       // idem
        bool disable_decoding_inf_mha = false;       // idem
        bool disable_decoding_inf_gqa = false;       // idem
        int32_t configure_decoding_attn_split_k = 1; // 0 off / 1 heuristic / 2 always
        int32_t specify_decoding_attn_tpb = 0;       // 0 heuristic / 256 / 512
        std::string quant_method = "none";           // ("online_i8i8" = W8A8 is not on the north-star path)
    };
    EngineConfig engine_config;
};

struct GeneratorConfig final {
    float top_p = 0.0f;
    int32_t top_k = 1;
    bool enable_penalty = false;
    int32_t max_running_batch = 0;
    int32_t max_input_tokens_per_request = 0;
    int32_t max_output_tokens_per_request = 0;
    int32_t max_total_tokens_per_request = 0;
    int32_t max_tokens_per_step = 0;
    std::set<int> stop_tokens;
    std::set<int> special_tokens;
    int max_cooldown_request = 2;
    bool enable_prefix_cache = false;
    int32_t max_prefill_batch = 0;
    bool enable_profiling = false;
};

struct ModelConfig final {
    int32_t hidden_dim = 0;
    int32_t intermediate_dim = 0;
    int32_t num_layers = 0;
    int32_t num_heads = 0;
    int32_t num_kv_heads = 0;
    int32_t vocab_size = 0;

    float norm_eps = 1e-5f;
    float rope_theta = 10000.f;
    int32_t max_position = 8192;

    int32_t cache_quant_bit = 0;
    int32_t cache_quant_group = 0;

    int32_t cache_layout = 0;
    int32_t cache_mode = 0;
    int32_t page_size = 0;

    int32_t weight_quant_bit = 0;      // optional key "weight_quant_bit": 0 / 8 (W8A16) / 4 (W4A16)
    int32_t weight_quant_group = 128;  // optional key "weight_quant_group"

    bool dynamic_batching = true;
    bool auto_causal = true;
};

bool ParseModelConfig(const std::string& model_param_path, ModelConfig* model_config);
bool ParseModelConfigFromString(const std::string& json_text, ModelConfig* model_config);

}}  // namespace ppl::llm
