// The per-step operator API between the scheduler and a backend (reference src/engine/llm_engine.h:40-149):
// ModelInput (what the generator packs), ModelOutput (what sampling returns) and LLMEngine::Execute
// = upload step inputs on every tensor-parallel rank -> run the decoder on every rank -> penalty -> sampling on rank 0.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "../common/config.h"
#include "../common/request.h"
#include "../common/resource.h"

namespace ppl { namespace llm {

struct ModelInput {
    int64_t decoding_batches = 0;
    int64_t max_seq_len = 0;
    int64_t max_kv_len = 0;
    int64_t max_pages = 0;

    std::vector<int64_t> token_inputs;
    std::vector<int64_t> seq_starts;
    std::vector<int64_t> start_pos;
    std::vector<int64_t> cache_indices;
    std::vector<int64_t> page_list;
    std::vector<int64_t> kv_starts;
    std::vector<float> temperatures;
    std::vector<float> top_p_list;
    std::vector<int32_t> top_k_list;

    std::vector<float> repetition_penalty_list;
    std::vector<float> presence_penalty_list;
    std::vector<float> frequency_penalty_list;
    std::vector<int64_t> batch_slots;
};

struct ModelOutput {
    std::vector<int32_t> output_token;
    std::vector<float> logprobs;
    void Clear() {
        output_token.clear();
        logprobs.clear();
    }
    void Resize(int32_t n) {
        output_token.resize(n);
        logprobs.resize(n);
    }
};

class LLMEngine final {
public:
    LLMEngine(const Resource& resource, const ModelConfig& model_config, bool enable_penalty, int32_t top_k, float top_p);

    ppl::common::RetCode Init(WorkerPerStepCounter* step_counter);
    ppl::common::RetCode Execute(const ModelInput& model_input, bool req_list_changed, bool is_prefix_cache_hit,
                                 ModelOutput* model_output, std::string* error_msg);

private:
    uint32_t tensor_parallel_size_;
    ppl::common::StaticThreadPool* device_worker_pool_;
    std::vector<Runtime*> runtimes_;
    uint64_t kv_cache_max_tokens_;
    PostProcessor* post_processor_;
    ModelConfig model_config_;
    bool enable_penalty_;
    int32_t top_k_;
    float top_p_;
    WorkerPerStepCounter* step_counter_ = nullptr;
};

}}  // namespace ppl::llm
