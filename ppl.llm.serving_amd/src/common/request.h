// Request / Response / Connection / profiler counters: the data the tools and the generator exchange
// (reference src/common/request.h:29-46, response.h:26-41, connection.h:28-35, profiler.h:27-115).
#pragma once
#include <stdint.h>

#include <memory>
#include <string>
#include <unordered_set>
#include <vector>

#include "ppl/common/retcode.h"

namespace ppl { namespace llm {

struct Request final {
    Request() {}
    Request(uint64_t _id, std::string _prompt, float _temperature, uint32_t _generation_length)
        : id(_id), prompt(std::move(_prompt)), temperature(_temperature), generation_length(_generation_length) {}
    uint64_t id = 0;
    std::string prompt;
    float temperature = 1.f;
    float top_p = 0.f;
    int32_t top_k = 1;
    float repetition_penalty = 1.f;
    float presence_penalty = 0.f;
    float frequency_penalty = 0.f;
    int32_t generation_length = 0;
    bool early_stopping = true;
    bool is_token_in_out = false;
    std::shared_ptr<std::vector<int>> token_ids;
    std::shared_ptr<std::unordered_set<int>> stop_tokens;
};

enum class FinishFlag { NOT_FINISHED, LENGTH, EOS_TOKEN, STOP_SEQUENCE };

struct Response final {
    uint64_t id = 0;
    std::string generated;
    int token = 0;
    FinishFlag finish_flag = FinishFlag::NOT_FINISHED;
    float logprob = 0.f;
    bool is_special = false;
};

struct GeneratorReqCounter final {
    uint64_t encode_cnt = 0;
    uint64_t encode_cost = 0;  // microseconds
    uint64_t output_tokens_per_req = 0;
    char padding[40];  // keep the producer-side and consumer-side counters on different cache lines
    uint64_t waiting_cnt = 0;
    uint64_t waiting_cost = 0;
};

struct WorkerPerStepCounter {
    struct Counters {
        uint64_t step_cnt = 0;
        uint64_t prepare_cost = 0;
        uint64_t set_input_cost = 0;
        uint64_t model_forward_cost = 0;
        uint64_t choose_token_cost = 0;  // penalty + sampling
        uint64_t post_process_cost = 0;
        uint64_t total_cost = 0;
        uint64_t input_token_cnt = 0;
        uint64_t output_token_cnt = 0;
        uint64_t cache_hit_count = 0;
    } global, current;
};

struct WorkerProfiler {
    uint64_t finished_task_cnt = 0;
    uint64_t kv_rest_blk = 0;
    uint64_t kv_max_blk = 0;
    uint64_t running_task = 0;
    uint64_t prefill_batch = 0;
    uint64_t prefill_tokens = 0;
    uint64_t max_running_task = 0;
    uint64_t pending_task_size = 0;
    uint64_t dev_mem_total = 0;
    uint64_t dev_mem_free = 0;
    WorkerPerStepCounter step_counter;
    GeneratorReqCounter req_counter;
};

void PrintProfiler(const WorkerProfiler& worker_profiler);

class Connection {
public:
    virtual ~Connection() {}
    virtual void OnProfiling(const std::shared_ptr<WorkerProfiler>&) = 0;
    virtual void OnTokenize(uint64_t id, const std::vector<int>&) = 0;
    virtual void Send(const std::vector<Response>&) = 0;
    virtual void NotifyFailure(uint64_t id, ppl::common::RetCode, const std::string& errmsg) = 0;
};

}}  // namespace ppl::llm
