// src/backends/hip: the backend plug-in that mirrors src/backends/cuda (reference resource_manager.h:85-123,
// resource_manager.cc:213-428, post_processor.h:27-69) on top of the C ABI of include/pplhip.h.  Thin adapters only:
// every device operation is a pplhip_* call.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "../../../../include/pplhip.h"
#include "../../common/config.h"
#include "../../common/resource.h"
#include "ppl/common/threadpool.h"

namespace ppl { namespace llm { namespace hip {

// ppl::nn::Runtime of one rank -> (pplhip_ctx, rank)
class HipRuntime final : public Runtime {
public:
    HipRuntime(pplhip_ctx* ctx, int rank) : ctx_(ctx), rank_(rank) {}
    ppl::common::RetCode SetInputs(const StepInputs&) override;
    ppl::common::RetCode Run(bool is_prefix_cache_hit) override;
    float* GetLogits(int64_t* batch_stride) override;
    const char* GetLastError() const override { return pplhip_last_error(ctx_, rank_); }

private:
    pplhip_ctx* ctx_;
    int rank_;
};

class HipPostProcessor final : public PostProcessor {
public:
    explicit HipPostProcessor(pplhip_ctx* ctx) : ctx_(ctx) {}
    ppl::common::RetCode InitPostProcessorMem(int max_running_batch, int vocab_size, bool enable_penalty) override;
    ppl::common::RetCode SampleTopKTopP(const float* logits_device, const float* temperatures_host, const int32_t* top_k_host,
                                        const float* top_p_host, int32_t batch, int32_t vocab_size, int32_t batch_stride,
                                        int32_t default_top_k, float default_top_p, bool req_list_changed,
                                        int32_t* output_host, float* logprob_host, bool enable_penalty) override;
    ppl::common::RetCode ApplyPenalty(const float* temperatures_host, const float* repetition_penalties_host,
                                      const float* presence_penalties_host, const float* frequency_penalties_host,
                                      const int64_t* batch_slots_host, const int64_t* token_inputs, const int64_t* seqstarts,
                                      const int64_t* start_pos, int32_t batch, int32_t vocab_size, bool req_list_changed,
                                      float* logits) override;

private:
    pplhip_ctx* ctx_;
};

// Owns everything device-side (context, KV slabs, runtimes, sampler, worker pool); LLMEngine / LLMGenerator hold
// non-owning pointers and must be destroyed first -- the ownership rule of the reference (resource_manager.h:86-109).
struct HipResourceManager final {
    ~HipResourceManager();
    ppl::common::RetCode Init(const ModelConfig& model_config, const ResourceConfig& resource_config);
    // fills a Resource the way tools/offline_inference.cc:367-373 does
    void FillResource(Resource* resource);

    ppl::common::StaticThreadPool device_worker_pool_;
    std::vector<ResourceItem> items;
    std::vector<std::unique_ptr<HipRuntime>> runtimes;
    std::unique_ptr<PostProcessor> post_processor;
    uint64_t kv_cache_max_tokens = 0;
    uint32_t tensor_parallel_size = 0;
    pplhip_ctx* ctx = nullptr;
};

}}}  // namespace ppl::llm::hip
