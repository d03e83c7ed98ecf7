// What the generator/engine require of a backend (reference: PostProcessor src/common/post_processor.h:25-43,
// Resource/ResourceItem src/common/resource.h:33-49, and the ppl.nn Runtime/Tensor surface the engine drives,
// src/engine/llm_engine.h:124-147, llm_engine.cc:29-116).  The reference binds eleven ppl::nn::Tensor objects by
// index; here the same data crosses the boundary as ONE StepInputs record per rank (-> pplhip_step), because the
// HIP runtime stages it in pinned memory and issues a single H2D copy.
#pragma once
#include <stdint.h>

#include <vector>

#include "ppl/common/retcode.h"
#include "ppl/common/threadpool.h"

namespace ppl { namespace llm {

class PostProcessor {
public:
    virtual ~PostProcessor() {}
    virtual ppl::common::RetCode InitPostProcessorMem(int max_running_batch, int vocab_size, bool enable_penalty) = 0;
    // logits_device -> output_host / logprob_host; blocking (the step's one synchronisation)
    virtual ppl::common::RetCode SampleTopKTopP(const float* logits_device, const float* temperatures_host,
                                                const int32_t* top_k_host, const float* top_p_host, int32_t batch,
                                                int32_t vocab_size, int32_t batch_stride, int32_t default_top_k,
                                                float default_top_p, bool req_list_changed, int32_t* output_host,
                                                float* logprob_host, bool enable_penalty) = 0;
    virtual ppl::common::RetCode ApplyPenalty(const float* temperatures_host, const float* repetition_penalties_host,
                                              const float* presence_penalties_host, const float* frequency_penalties_host,
                                              const int64_t* batch_slots_host, const int64_t* token_inputs,
                                              const int64_t* seqstarts, const int64_t* start_pos, int32_t batch,
                                              int32_t vocab_size, bool req_list_changed, float* logits) = 0;
};

// one step of one rank: the 11-input contract of src/engine/llm_engine.h:124-138 (attn_mask is never written by the
// reference, llm_engine.cc:29-111, and is not carried)
struct StepInputs {
    int64_t batch = 0, num_tokens = 0;
    int64_t decoding_batches = 0, max_seq_len = 0, max_kv_len = 0, max_pages = 0;
    const int64_t* token_inputs = nullptr;
    const int64_t* seq_starts = nullptr;
    const int64_t* kv_starts = nullptr;
    const int64_t* start_pos = nullptr;
    const int64_t* cache_indices = nullptr;  // mode 0: [B]; mode 1: [B, max_pages], only read when req_list_changed
    bool req_list_changed = true;
};

// the runtime handle of one tensor-parallel rank (ppl::nn::Runtime + its bound tensors in the reference)
class Runtime {
public:
    virtual ~Runtime() {}
    virtual ppl::common::RetCode SetInputs(const StepInputs&) = 0;           // SetInputTask (async)
    virtual ppl::common::RetCode Run(bool is_prefix_cache_hit) = 0;          // RunModelTask (async)
    virtual float* GetLogits(int64_t* batch_stride) = 0;                     // logits tensor: device ptr + dim(1)
    // device-resident step arrays for ApplyPenalty (llm_engine.cc:207-211)
    virtual const int64_t* GetTokenInputsDevice() const { return nullptr; }
    virtual const int64_t* GetSeqStartsDevice() const { return nullptr; }
    virtual const int64_t* GetStartPosDevice() const { return nullptr; }
    virtual const char* GetLastError() const { return ""; }
};

struct ResourceItem final {
    void* kv_cache_mem = nullptr;
    void* kv_scale_mem = nullptr;
    Runtime* runtime = nullptr;
};

class Tokenizer;  // out of scope (token-in/token-out path only); kept so that Resource has the reference's shape

struct Resource final {
    uint32_t tensor_parallel_size = 0;
    uint64_t kv_cache_max_tokens = 0;
    std::vector<ResourceItem> items;
    PostProcessor* post_processor = nullptr;
    ppl::common::StaticThreadPool* device_worker_pool_ = nullptr;
    const Tokenizer* tokenizer = nullptr;
};

}}  // namespace ppl::llm
