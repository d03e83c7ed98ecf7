// Continuous-batching scheduler of the hot path (reference src/generator/llm_generator.h:127-182,
// llm_generator.cc:574-786): request queue -> admission (token budget, length clamps, KV reservation, prefix-cache
// lookup) -> ModelInput packing -> LLMEngine::Execute -> finish detection -> asynchronous send -> KV release and
// batch compaction.  Public surface (Init / Process / ClearTask / GetPendingTaskNum) is the reference's.
#pragma once
#include <pthread.h>

#include <atomic>
#include <chrono>
#include <map>
#include <memory>
#include <unordered_set>
#include <vector>

#include "../common/config.h"
#include "../common/request.h"
#include "../common/resource.h"
#include "../engine/llm_engine.h"
#include "../tokenizer/tokenizer.h"
#include "../utils/index_manager.h"
#include "../utils/mpsc_request_scheduler.h"
#include "../utils/prefix_cache_manager.h"
#include "ppl/common/allocators.h"
#include "ppl/common/mpsc_queue.h"
#include "ppl/common/threadpool.h"

namespace ppl { namespace llm {

// one generated token on its way to the connection
struct TidGenToken final {
    uint64_t tid;
    int token;
    float logprob;
    FinishFlag finish_flag;
    uint64_t steps;
    bool is_token_in_out;
    bool is_special;
};

struct FinishedTaskInfo final {
    enum { UNKNOWN, FROM_WORKER, FROM_CONN };
    FinishedTaskInfo(uint64_t fid = UINT64_MAX, uint32_t ftype = UNKNOWN) : id(fid), type(ftype) {}
    uint64_t id;
    uint32_t type;
};

// per running request (reference TidData, llm_generator.h:79-103)
struct TidData final {
    uint64_t tid = 0;
    float temperature = 1.f;
    float top_p = 0.f;
    int32_t top_k = 1;
    float repetition_penalty = 1.f;
    float presence_penalty = 0.f;
    float frequency_penalty = 0.f;
    bool early_stopping = true;
    int32_t rest_iters = 0;
    bool is_token_in_out = false;
    int64_t total_len = 0;  // prompt + rest_iters at admission; KV reserved = total_len - 1
    std::shared_ptr<std::unordered_set<int>> stop_tokens;
    std::shared_ptr<std::vector<int>> next_tokens;
    int64_t start_pos = 0;
    uint64_t cache_index = 0;
    std::vector<int64_t> page_list;
    int64_t slot_index = 0;
    int32_t steps = 0;
    int32_t gen_tokens_cnt = 0;
    std::vector<uint64_t> hash_list;  // hashes of this request's full prompt pages (cached + newly inserted)
    int64_t cache_hit_count = 0;
};

struct LlmRequest final : public ppl::common::MPSCQueue::Node {
    std::shared_ptr<Request> orig;
    std::chrono::time_point<std::chrono::high_resolution_clock> enqueue_ts;
};

// test hook: receives every ModelInput right before Execute
typedef void (*StepObserver)(void* arg, uint64_t step, const ModelInput& input, bool req_list_changed, bool is_prefix_cache_hit);

class LLMGenerator final {
public:
    LLMGenerator(const Resource& resource, const GeneratorConfig& generator_config, const ModelConfig& model_config,
                 Connection* conn);
    ~LLMGenerator();

    ppl::common::RetCode Init();
    void Process(const std::shared_ptr<Request>&);
    void ClearTask(uint64_t tid) { finished_tasks_.Push(FinishedTaskInfo(tid, FinishedTaskInfo::FROM_CONN)); }
    uint32_t GetPendingTaskNum() const { return sched_.GetPendingSize(); }

    void SetStepObserver(StepObserver f, void* arg) { observer_ = f; observer_arg_ = arg; }
    const WorkerProfiler& GetProfiler() const { return *worker_profiler_; }
    // nothing queued and no batch running (tools use it to drain before shutdown)
    bool IsIdle() const { return sched_.GetPendingSize() == 0 && !generating_.load(std::memory_order_acquire); }

private:
    struct Admission;  // scratch of one admission check (RequestCheckResult in the reference)

    ppl::common::RetCode CheckParameters() const;
    void Generate();
    bool AdmitRequest(const LlmRequest& req, Admission* adm, int32_t* cool_down, bool* is_prefix_cache_hit);
    bool ReserveKv(const LlmRequest& req, Admission* adm, int32_t* cool_down, bool* is_prefix_cache_hit);
    bool StartRequest(const LlmRequest& req, const Admission& adm, ModelInput* model_input);
    void PackStep(bool req_list_changed, ModelInput* model_input) const;
    void DeleteTasks(ModelInput* model_input);
    void CompactBatch(ModelInput* model_input);
    void ReleaseResource();
    void SendTokens(const std::vector<TidGenToken>& tokens);
    static void* GeneratorThreadFunc(void*);

private:
    const Tokenizer* tokenizer_;
    GeneratorConfig generator_config_;
    ModelConfig model_config_;
    Connection* conn_;
    uint64_t kv_cache_max_tokens_ = 0;
    LLMEngine llm_engine_;
    ppl::common::StaticThreadPool decoder_thread_pool_;

    // running batch: row order == ModelInput row order
    std::vector<TidData*> tid_list_;
    std::map<uint64_t, TidData> tid_data_map_;
    bool req_list_changed_ = true;
    ppl::common::TypedMPSCQueue<FinishedTaskInfo> finished_tasks_;

    utils::IndexManager idx_mgr_;
    utils::IndexManager batch_slots_mgr_;
    ppl::common::PageManager page_mgr_;
    utils::PrefixCacheManager prefix_cache_mgr_;
    std::shared_ptr<WorkerProfiler> worker_profiler_;

    std::atomic<bool> generate_thread_active_{false};
    std::atomic<bool> generating_{false};
    pthread_t generate_thread_;
    ppl::common::EventCount req_signal_;
    utils::MPSCRequestScheduler<LlmRequest> sched_;

    // U+FFFD buffering of the text path (llm_generator.cc:84-99)
    std::map<uint64_t, std::vector<int>> decode_buffer_;

    StepObserver observer_ = nullptr;
    void* observer_arg_ = nullptr;
    static constexpr int DECODER_THREAD_NUM = 1;
};

}}  // namespace ppl::llm
