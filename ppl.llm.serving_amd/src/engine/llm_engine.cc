#include "llm_engine.h"

#include "../utils/utils.h"
#include "ppl/common/log.h"

using namespace ppl::common;

namespace ppl { namespace llm {

LLMEngine::LLMEngine(const Resource& resource, const ModelConfig& model_config, bool enable_penalty, int32_t top_k, float top_p)
    : tensor_parallel_size_(resource.tensor_parallel_size)
    , device_worker_pool_(resource.device_worker_pool_)
    , kv_cache_max_tokens_(resource.kv_cache_max_tokens)
    , post_processor_(resource.post_processor)
    , model_config_(model_config)
    , enable_penalty_(enable_penalty)
    , top_k_(top_k)
    , top_p_(top_p) {
    for (uint32_t i = 0; i < tensor_parallel_size_; ++i) runtimes_.push_back(resource.items[i].runtime);
}

// The reference's Init reshapes the kv_cache / kv_scale tensors per cache_layout (src/engine/llm_engine.cc:118-169).
// In this build the slab shape is a property of the backend (it derives the four layouts' strides itself), so Init
// only validates what the reference validates there.
RetCode LLMEngine::Init(WorkerPerStepCounter* step_counter) {
    step_counter_ = step_counter;
    if (model_config_.cache_layout < 0 || model_config_.cache_layout > 3) {
        LOG(ERROR) << "impossible status: cache_layout = [" << model_config_.cache_layout << "]";
        return RC_INVALID_VALUE;
    }
    for (auto* rt : runtimes_) {
        if (!rt) {
            LOG(ERROR) << "resource item without runtime";
            return RC_INVALID_VALUE;
        }
    }
    return RC_SUCCESS;
}

static RetCode UploadInputs(uint32_t rank, const StepInputs& in, const std::vector<Runtime*>& runtimes) {
    const RetCode rc = runtimes[rank]->SetInputs(in);
    if (rc != RC_SUCCESS) LOG(ERROR) << "set inputs on rank [" << rank << "] failed: " << runtimes[rank]->GetLastError();
    return rc;
}

static RetCode RunDecoder(uint32_t rank, bool is_prefix_cache_hit, const std::vector<Runtime*>& runtimes) {
    const RetCode rc = runtimes[rank]->Run(is_prefix_cache_hit);
    if (rc != RC_SUCCESS) LOG(ERROR) << "run on rank [" << rank << "] failed: " << runtimes[rank]->GetLastError();
    return rc;
}

RetCode LLMEngine::Execute(const ModelInput& in, bool req_list_changed, bool is_prefix_cache_hit, ModelOutput* out,
                           std::string* error_msg) {
    const int32_t running_batch = (int32_t)in.start_pos.size();
    RetCode rc;

    StepInputs step;
    step.batch = running_batch;
    step.num_tokens = (int64_t)in.token_inputs.size();
    step.decoding_batches = in.decoding_batches;
    step.max_seq_len = in.max_seq_len;
    step.max_kv_len = in.max_kv_len;
    step.max_pages = in.max_pages;
    step.token_inputs = in.token_inputs.data();
    step.seq_starts = in.seq_starts.data();
    step.kv_starts = in.kv_starts.data();
    step.start_pos = in.start_pos.data();
    // cache_mode 0: cache_indices every step; cache_mode 1: the page table, consumed only when the batch changed
    // (src/engine/llm_engine.cc:63-71)
    step.cache_indices = model_config_.cache_mode == 0 ? in.cache_indices.data() : in.page_list.data();
    step.req_list_changed = req_list_changed;

    {
        utils::TimingGuard timing(&step_counter_->current.set_input_cost);
        rc = utils::ParallelExecute(UploadInputs, device_worker_pool_, step, runtimes_);
        if (rc != RC_SUCCESS) {
            *error_msg = "ParallelExecute(SetInputTask) failed: " + std::string(GetRetCodeStr(rc));
            LOG(ERROR) << *error_msg;
            return RC_OTHER_ERROR;
        }
    }
    step_counter_->global.set_input_cost += step_counter_->current.set_input_cost;

    {
        utils::TimingGuard timing(&step_counter_->current.model_forward_cost);
        rc = utils::ParallelExecute(RunDecoder, device_worker_pool_, is_prefix_cache_hit, runtimes_);
        if (rc != RC_SUCCESS) {
            *error_msg = "ParallelExecute(RunModelTask) failed: " + std::string(GetRetCodeStr(rc));
            LOG(ERROR) << *error_msg;
            return RC_OTHER_ERROR;
        }
    }
    step_counter_->global.model_forward_cost += step_counter_->current.model_forward_cost;

    {
        utils::TimingGuard timing(&step_counter_->current.choose_token_cost);
        Runtime* rt0 = runtimes_[0];  // sampling happens on rank 0's logits only (src/engine/llm_engine.cc:200)
        int64_t stride = 0;
        float* logits = rt0->GetLogits(&stride);
        if (enable_penalty_) {
            rc = post_processor_->ApplyPenalty(in.temperatures.data(), in.repetition_penalty_list.data(), nullptr, nullptr,
                                               in.batch_slots.data(), rt0->GetTokenInputsDevice(), rt0->GetSeqStartsDevice(),
                                               rt0->GetStartPosDevice(), running_batch, model_config_.vocab_size,
                                               req_list_changed, logits);
            if (rc != RC_SUCCESS) {
                *error_msg = "Apply Penalty failed: " + std::string(GetRetCodeStr(rc));
                LOG(ERROR) << *error_msg;
                return RC_OTHER_ERROR;
            }
        }
        // only top_k_list[0] reaches the kernel (src/engine/llm_engine.cc:219)
        const int32_t default_top_k = in.top_k_list.empty() ? top_k_ : in.top_k_list[0];
        rc = post_processor_->SampleTopKTopP(logits, in.temperatures.data(), in.top_k_list.data(), in.top_p_list.data(),
                                             running_batch, model_config_.vocab_size, (int32_t)stride, default_top_k, top_p_,
                                             req_list_changed, out->output_token.data(), out->logprobs.data(),
                                             enable_penalty_);
        if (rc != RC_SUCCESS) {
            *error_msg = "SampleTopKTopP failed: " + std::string(GetRetCodeStr(rc));
            LOG(ERROR) << *error_msg;
            return RC_OTHER_ERROR;
        }
    }
    step_counter_->global.choose_token_cost += step_counter_->current.choose_token_cost;
    step_counter_->current.output_token_cnt = running_batch;
    step_counter_->global.output_token_cnt += running_batch;
    return RC_SUCCESS;
}

}}  // namespace ppl::llm
