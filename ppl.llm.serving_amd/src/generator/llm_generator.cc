#include "llm_generator.h"

#include <math.h>
#include <string.h>

#include <algorithm>
#include <limits>

#include "../utils/utils.h"
#include "ppl/common/log.h"

using namespace ppl::common;

namespace ppl { namespace llm {

// scratch of one admission check (reference RequestCheckResult, llm_generator.cc:31-43)
struct LLMGenerator::Admission {
    int64_t cache_index = INT64_MAX;
    std::vector<int64_t> page_list;
    int64_t slot_index = INT64_MAX;
    int rest_iters = -1;
    int first_fill_len = 0;
    int total_tokens_per_step = 0;
    std::vector<uint64_t> hash_list;
    int64_t cache_hit_count = 0;
    int32_t running_batch = 0;
    int32_t prefill_batch = 0;
    std::string errmsg;

    void ResetForRequest(int prompt_len) {
        cache_index = INT64_MAX;
        page_list.clear();
        slot_index = INT64_MAX;
        rest_iters = -1;
        first_fill_len = prompt_len;
        hash_list.clear();
        cache_hit_count = 0;
        errmsg.clear();
    }
};

LLMGenerator::LLMGenerator(const Resource& resource, const GeneratorConfig& generator_config, const ModelConfig& model_config,
                           Connection* conn)
    : tokenizer_(resource.tokenizer)
    , generator_config_(generator_config)
    , model_config_(model_config)
    , conn_(conn)
    , kv_cache_max_tokens_(resource.kv_cache_max_tokens)
    , llm_engine_(resource, model_config, generator_config.enable_penalty, generator_config.top_k, generator_config.top_p) {
    idx_mgr_.Init(kv_cache_max_tokens_);
    batch_slots_mgr_.Init(generator_config.max_running_batch);
    page_mgr_.Init(kv_cache_max_tokens_, model_config_.page_size);
    worker_profiler_ = std::make_shared<WorkerProfiler>();
}

LLMGenerator::~LLMGenerator() {
    if (generate_thread_active_.load(std::memory_order_relaxed)) {
        generate_thread_active_.store(false, std::memory_order_release);
        req_signal_.NotifyOne();
        pthread_join(generate_thread_, nullptr);
    }
}

// reference CheckParameters, llm_generator.cc:114-144 (accepted combinations: SURVEY.md Q9)
RetCode LLMGenerator::CheckParameters() const {
    const ModelConfig& m = model_config_;
    if (!m.auto_causal) { LOG(ERROR) << "only support auto_causal == true"; return RC_INVALID_VALUE; }
    if (m.cache_mode != 0 && m.cache_mode != 1) { LOG(ERROR) << "unsupported cache_mode: " << m.cache_mode; return RC_INVALID_VALUE; }
    if (m.cache_layout < 0 || m.cache_layout > 3) { LOG(ERROR) << "only support cache_layout 0..3"; return RC_INVALID_VALUE; }
    const bool int8_kv = m.cache_quant_bit == 8 && m.cache_quant_group == 8;
    const bool fp16_kv = m.cache_quant_bit == 0 && m.cache_quant_group == 1;
    if (!int8_kv && !fp16_kv) {
        LOG(ERROR) << "only support (cache_quant_bit == 8 and cache_quant_group == 8) or (cache_quant_bit == 0 and cache_quant_group == 1)";
        return RC_INVALID_VALUE;
    }
    if (!m.dynamic_batching) { LOG(ERROR) << "only support dynamic_batching == true"; return RC_INVALID_VALUE; }
    if (m.cache_mode == 1 && m.page_size <= 0) { LOG(ERROR) << "cache_mode 1 needs page_size > 0"; return RC_INVALID_VALUE; }
    return RC_SUCCESS;
}

RetCode LLMGenerator::Init() {
    RetCode rc = CheckParameters();
    if (rc != RC_SUCCESS) { LOG(ERROR) << "CheckParameters failed."; return rc; }
    rc = llm_engine_.Init(&worker_profiler_->step_counter);
    if (rc != RC_SUCCESS) { LOG(ERROR) << "LLM Engine Init failed."; return rc; }
    // the rope table (and pplhip_set_inputs' range check) ends at max_position: a request may never grow past it, or one
    // long request would fail Execute for the whole running batch.  Clamp per request at admission instead.
    if (model_config_.max_position > 0 && generator_config_.max_total_tokens_per_request > model_config_.max_position) {
        LOG(WARNING) << "max_total_tokens_per_request [" << generator_config_.max_total_tokens_per_request
                     << "] > model max_position [" << model_config_.max_position << "]. use [" << model_config_.max_position << "]";
        generator_config_.max_total_tokens_per_request = model_config_.max_position;
    }
    rc = decoder_thread_pool_.Init(DECODER_THREAD_NUM);
    if (rc != RC_SUCCESS) { LOG(ERROR) << "Init decoder thread pool error"; return RC_OTHER_ERROR; }
    generate_thread_active_.store(true, std::memory_order_release);
    if (pthread_create(&generate_thread_, nullptr, GeneratorThreadFunc, this) != 0) {
        generate_thread_active_.store(false, std::memory_order_relaxed);
        LOG(ERROR) << "create generator thread failed.";
        return RC_OTHER_ERROR;
    }
    return RC_SUCCESS;
}

// reference Process, llm_generator.cc:788-814
void LLMGenerator::Process(const std::shared_ptr<Request>& req) {
    uint64_t encode_cost = 0;
    if (req->token_ids) req->is_token_in_out = true;
    {
        utils::TimingGuard timing(&encode_cost);
        if (!req->is_token_in_out) {
            if (!tokenizer_) {
                conn_->NotifyFailure(req->id, RC_UNSUPPORTED, "no tokenizer: only token-in/token-out requests are accepted");
                return;
            }
            req->token_ids = std::make_shared<std::vector<int>>();
            tokenizer_->Encode(req->prompt.data(), (uint32_t)req->prompt.size(), req->token_ids.get());
            req->stop_tokens = std::make_shared<std::unordered_set<int>>();
            req->stop_tokens->insert(tokenizer_->GetEosId());
            conn_->OnTokenize(req->id, *req->token_ids);
        }
    }
    worker_profiler_->step_counter.global.input_token_cnt += req->token_ids->size();
    ++worker_profiler_->req_counter.encode_cnt;
    worker_profiler_->req_counter.encode_cost += encode_cost;

    auto* lreq = new LlmRequest();
    lreq->orig = req;
    lreq->enqueue_ts = std::chrono::high_resolution_clock::now();
    if (sched_.PushRequest(lreq)) req_signal_.NotifyOne();
}

// reference GeneratorThreadFunc, llm_generator.cc:342-366
void* LLMGenerator::GeneratorThreadFunc(void* arg) {
    auto* g = static_cast<LLMGenerator*>(arg);
    while (true) {
        while (true) {
            const auto key = g->req_signal_.PrepareWait();
            if (!g->generate_thread_active_.load(std::memory_order_acquire)) {
                g->req_signal_.CancelWait();
                return nullptr;
            }
            if (g->sched_.GetPendingSize() > 0) {
                g->generating_.store(true, std::memory_order_release);
                g->req_signal_.CancelWait();
                break;
            }
            LOG(INFO) << "waiting for request ...";
            g->req_signal_.CommitWait(key);
        }
        g->generating_.store(true, std::memory_order_release);
        g->Generate();
        g->generating_.store(false, std::memory_order_release);
    }
    return nullptr;
}

// ------------------------------------------------------------------------------------------------ admission

// length clamps: reference CheckTotalLen, llm_generator.cc:441-478 (including that the total-length clamp is
// computed from the REQUESTED generation length and may exceed max_output_tokens_per_request)
static bool ClampLengths(const GeneratorConfig& cfg, const Request& r, int* first_fill_len, int* rest_iters, std::string* errmsg) {
    const std::string idstr = "id [" + std::to_string(r.id) + "]";
    if (*first_fill_len > cfg.max_input_tokens_per_request) {
        *errmsg = idstr + " invalid input token len: " + std::to_string(*first_fill_len) +
                  ", server allowed max input len: " + std::to_string(cfg.max_input_tokens_per_request);
        *first_fill_len = -1;
        return false;
    }
    *rest_iters = r.generation_length;
    if (r.generation_length > cfg.max_output_tokens_per_request) {
        const std::string msg = idstr + ": generation len in request is [" + std::to_string(r.generation_length) + "] > [" +
                                std::to_string(cfg.max_output_tokens_per_request) + "] from cmd. use [" +
                                std::to_string(cfg.max_output_tokens_per_request) + "]";
        LOG(WARNING) << msg;
        *rest_iters = cfg.max_output_tokens_per_request;
        if (*rest_iters <= 0) { *errmsg = msg; return false; }
    }
    if (*first_fill_len + r.generation_length > cfg.max_total_tokens_per_request) {
        const int clamped = cfg.max_total_tokens_per_request - *first_fill_len;
        const std::string msg = idstr + ": total len in request is [" + std::to_string(*first_fill_len + r.generation_length) +
                                "] > [" + std::to_string(cfg.max_total_tokens_per_request) + "] from cmd. use [" +
                                std::to_string(clamped) + "]";
        LOG(WARNING) << msg;
        *rest_iters = clamped;
        if (*rest_iters <= 0) { *errmsg = msg; return false; }
    }
    return true;
}

// KV reservation for the request's whole lifetime, total_len = prompt + rest_iters - 1 tokens, up front
// (reference CheckAndAllocGPUMemory, llm_generator.cc:480-572)
bool LLMGenerator::ReserveKv(const LlmRequest& req, Admission* adm, int32_t* cool_down, bool* is_prefix_cache_hit) {
    const uint64_t total_len = (uint64_t)adm->first_fill_len + adm->rest_iters - 1;
    if (model_config_.cache_mode == 0) {
        adm->cache_index = idx_mgr_.Alloc(total_len);
        if (adm->cache_index == INT64_MAX) {
            // wait until a few running requests finish before trying again (llm_generator.cc:488-492)
            const int running = (int)tid_list_.size();
            *cool_down = std::min(std::max(1, (int)floorf(running * 0.1f)), generator_config_.max_cooldown_request);
            return false;
        }
    } else if (generator_config_.enable_prefix_cache) {
        const std::vector<int>& tokens = *req.orig->token_ids;
        const int64_t P = model_config_.page_size;
        // 1. longest chain of cached full pages: h_i = HashCombine(h_{i-1}, page i tokens)
        uint64_t prev_hash = 0, start = 0;
        for (; start + P <= tokens.size(); start += P) {
            const uint64_t h = utils::HashCombine(prev_hash, tokens.data() + start, (int32_t)P);
            const int64_t page_id = prefix_cache_mgr_.Find(h);
            if (page_id == -1) break;
            prev_hash = h;
            adm->page_list.push_back(page_id);
            adm->hash_list.push_back(h);
        }
        prefix_cache_mgr_.IncRefCount(adm->hash_list.data(), (int64_t)adm->hash_list.size());
        // 2. pages still needed; evict unreferenced cached pages if the pool is short
        const int64_t avail = page_mgr_.GetAvail();
        const int64_t need = ((int64_t)total_len - (int64_t)start + P - 1) / P;
        if (avail < need) {
            std::vector<int64_t> evicted;
            prefix_cache_mgr_.Evict(need - avail, &evicted);
            page_mgr_.Free(evicted.data(), (int64_t)evicted.size());
            if ((int64_t)evicted.size() < need - avail) {
                prefix_cache_mgr_.DecRefCount(adm->hash_list.data(), (int64_t)adm->hash_list.size());
                return false;
            }
        }
        adm->cache_hit_count = (int64_t)adm->hash_list.size() * P;
        worker_profiler_->step_counter.global.cache_hit_count += adm->cache_hit_count;
        if (adm->cache_hit_count != 0) {
            *is_prefix_cache_hit = true;
            LOG(INFO) << "Cache Hit [" << adm->cache_hit_count << "]/[" << tokens.size() << "] input tokens";
        }
        if (page_mgr_.Alloc(need, &adm->page_list) != RC_SUCCESS) {
            LOG(WARNING) << "page alloc failed after eviction";
            prefix_cache_mgr_.DecRefCount(adm->hash_list.data(), (int64_t)adm->hash_list.size());
            return false;
        }
        // 3. publish the remaining full prompt pages
        for (uint64_t pos = start; pos + P <= tokens.size(); pos += P) {
            const uint64_t h = utils::HashCombine(prev_hash, tokens.data() + pos, (int32_t)P);
            prefix_cache_mgr_.Insert(h, adm->page_list[pos / P]);
            prev_hash = h;
            adm->hash_list.push_back(h);
        }
    } else {
        const int64_t pages = ((int64_t)total_len + model_config_.page_size - 1) / model_config_.page_size;
        if (page_mgr_.Alloc(pages, &adm->page_list) != RC_SUCCESS) return false;
    }
    if (generator_config_.enable_penalty) {
        adm->slot_index = batch_slots_mgr_.Alloc(1);
        if (adm->slot_index == INT64_MAX) {
            LOG(ERROR) << "alloc batch slot error, available [" << batch_slots_mgr_.GetAvailableBlockNum() << "]";
            return false;
        }
    }
    return true;
}

// the admission predicate handed to the request scheduler (reference check_func, llm_generator.cc:590-617).
// true  -> the request leaves the queue (either admitted, or invalid and failed by StartRequest)
// false -> it stays at the head (stash) and admission stops for this step
bool LLMGenerator::AdmitRequest(const LlmRequest& req, Admission* adm, int32_t* cool_down, bool* is_prefix_cache_hit) {
    adm->ResetForRequest((int)req.orig->token_ids->size());
    // the prompt length counts against the step budget BEFORE the check and even for prefix-cache hits
    adm->total_tokens_per_step += adm->first_fill_len;
    if (adm->total_tokens_per_step > generator_config_.max_tokens_per_step) return false;
    if (adm->first_fill_len == 0) {  // deviation: an empty prompt would reserve zero KV slots and feed no token
        adm->errmsg = "id [" + std::to_string(req.orig->id) + "] empty prompt";
        adm->first_fill_len = -1;
        return true;
    }
    if (!ClampLengths(generator_config_, *req.orig, &adm->first_fill_len, &adm->rest_iters, &adm->errmsg)) {
        LOG(ERROR) << adm->errmsg;
        return true;
    }
    if (adm->rest_iters <= 0) {  // deviation: the reference reserves KV here and leaks it when StartRequest rejects
        adm->errmsg = "id [" + std::to_string(req.orig->id) + "] generation length <= 0";
        return true;
    }
    if (!ReserveKv(req, adm, cool_down, is_prefix_cache_hit)) return false;
    ++adm->running_batch;
    ++adm->prefill_batch;
    return true;
}

// reference ParseRequest, llm_generator.cc:193-261
bool LLMGenerator::StartRequest(const LlmRequest& req, const Admission& adm, ModelInput* in) {
    const Request& r = *req.orig;
    if (adm.rest_iters <= 0 || adm.first_fill_len == -1) {
        conn_->NotifyFailure(r.id, RC_INVALID_VALUE, adm.errmsg);
        return true;
    }
    const int mode = model_config_.cache_mode;
    if ((mode == 0 && adm.cache_index == INT64_MAX) || (mode == 1 && adm.page_list.empty())) {
        LOG(ERROR) << "catch invalid cache_index or page list";
        return false;
    }
    TidData& t = tid_data_map_.emplace(r.id, TidData()).first->second;
    t.tid = r.id;
    t.temperature = r.temperature;
    t.top_p = r.top_p;
    t.top_k = r.top_k;
    t.repetition_penalty = r.repetition_penalty;
    t.presence_penalty = r.presence_penalty;
    t.frequency_penalty = r.frequency_penalty;
    t.early_stopping = r.early_stopping;
    t.rest_iters = adm.rest_iters;
    t.total_len = adm.first_fill_len + adm.rest_iters;
    t.stop_tokens = r.stop_tokens;
    t.is_token_in_out = r.is_token_in_out;
    t.slot_index = adm.slot_index;
    if (mode == 0) {
        t.cache_index = (uint64_t)adm.cache_index;
    } else {
        t.page_list = adm.page_list;
        t.hash_list = adm.hash_list;
        t.cache_hit_count = adm.cache_hit_count;
    }
    // where the prefill starts: no hit -> 0; whole prompt cached -> recompute only its last token; else -> the hit
    const int64_t hit = adm.cache_hit_count;
    if (hit == 0) {
        t.next_tokens = r.token_ids;
        t.start_pos = 0;
    } else if ((size_t)hit == r.token_ids->size()) {
        t.next_tokens = std::make_shared<std::vector<int>>(1, r.token_ids->back());
        t.start_pos = hit - 1;
    } else {
        t.next_tokens = std::make_shared<std::vector<int>>(r.token_ids->begin() + hit, r.token_ids->end());
        t.start_pos = hit;
    }
    tid_list_.push_back(&t);
    in->start_pos.push_back(t.start_pos);
    in->temperatures.push_back(t.temperature);
    in->top_p_list.push_back(t.top_p);
    in->top_k_list.push_back(t.top_k);
    in->repetition_penalty_list.push_back(t.repetition_penalty);
    in->presence_penalty_list.push_back(t.presence_penalty);
    in->frequency_penalty_list.push_back(t.frequency_penalty);
    in->batch_slots.push_back(t.slot_index);
    if (mode == 0) in->cache_indices.push_back(adm.cache_index);
    else in->max_pages = std::max<int64_t>((int64_t)t.page_list.size(), in->max_pages);
    return true;
}

// ------------------------------------------------------------------------------------------------ packing

// reference UpdateInput, llm_generator.cc:263-298
void LLMGenerator::PackStep(bool req_list_changed, ModelInput* in) const {
    const size_t n = tid_list_.size();
    in->max_seq_len = 0;
    in->max_kv_len = 0;
    in->token_inputs.clear();
    in->seq_starts.assign(1, 0);
    in->kv_starts.assign(1, 0);
    in->seq_starts.reserve(n + 1);
    in->kv_starts.reserve(n + 1);
    const bool repack_pages = req_list_changed && model_config_.cache_mode == 1;
    if (repack_pages) in->page_list.assign(n * in->max_pages, INT64_MAX);
    for (size_t i = 0; i < n; ++i) {
        const TidData* t = tid_list_[i];
        const int64_t seqlen = (int64_t)t->next_tokens->size();
        in->token_inputs.insert(in->token_inputs.end(), t->next_tokens->begin(), t->next_tokens->end());
        in->seq_starts.push_back(in->seq_starts[i] + seqlen);
        in->kv_starts.push_back(in->kv_starts[i] + t->start_pos + seqlen);
        in->max_seq_len = std::max(in->max_seq_len, seqlen);
        in->max_kv_len = std::max(in->max_kv_len, t->start_pos + seqlen);
        if (repack_pages) std::copy(t->page_list.begin(), t->page_list.end(), in->page_list.begin() + i * in->max_pages);
    }
}

// reference RemoveFinishedTask, llm_generator.cc:300-340: stable compaction of every per-row vector
void LLMGenerator::CompactBatch(ModelInput* in) {
    const int mode = model_config_.cache_mode;
    size_t keep = 0;
    if (mode == 1) in->max_pages = 0;
    for (size_t i = 0; i < tid_list_.size(); ++i) {
        if (!tid_list_[i]) continue;
        tid_list_[keep] = tid_list_[i];
        if (mode == 0) in->cache_indices[keep] = in->cache_indices[i];
        else in->max_pages = std::max<int64_t>(in->max_pages, (int64_t)tid_list_[i]->page_list.size());
        in->start_pos[keep] = in->start_pos[i];
        in->temperatures[keep] = in->temperatures[i];
        in->top_p_list[keep] = in->top_p_list[i];
        in->top_k_list[keep] = in->top_k_list[i];
        in->repetition_penalty_list[keep] = in->repetition_penalty_list[i];
        in->presence_penalty_list[keep] = in->presence_penalty_list[i];
        in->frequency_penalty_list[keep] = in->frequency_penalty_list[i];
        in->batch_slots[keep] = in->batch_slots[i];
        ++keep;
    }
    tid_list_.resize(keep);
    if (mode == 0) in->cache_indices.resize(keep);
    in->start_pos.resize(keep);
    in->temperatures.resize(keep);
    in->top_p_list.resize(keep);
    in->top_k_list.resize(keep);
    in->repetition_penalty_list.resize(keep);
    in->presence_penalty_list.resize(keep);
    in->frequency_penalty_list.resize(keep);
    in->batch_slots.resize(keep);
    LOG(DEBUG) << "Rest tasks: " << keep;
}

// reference DeleteTasks, llm_generator.cc:387-439
void LLMGenerator::DeleteTasks(ModelInput* in) {
    FinishedTaskInfo info;
    while (finished_tasks_.Pop(&info)) {
        auto it = tid_data_map_.find(info.id);
        if (it == tid_data_map_.end()) continue;  // finished by the worker and cancelled by the connection at once
        TidData& t = it->second;
        --in->decoding_batches;
        size_t row = 0;
        while (row < tid_list_.size() && !(tid_list_[row] && tid_list_[row]->tid == info.id)) ++row;
        if (row == tid_list_.size()) continue;
        tid_list_[row] = nullptr;
        // the batch rows shift: the device page table, the penalty slots and the sampler's per-row parameters must be
        // re-uploaded by the next Execute even when nothing finished in the worker and nothing new was admitted (a
        // cancel from the connection on an otherwise quiet step; the reference leaves the flag untouched here)
        req_list_changed_ = true;
        if (model_config_.cache_mode == 0) {
            idx_mgr_.Free(t.cache_index, (uint64_t)t.total_len - 1);
        } else if (generator_config_.enable_prefix_cache) {
            // hashed prompt pages go back to the prefix cache (LRU once unreferenced); the rest to the page pool
            const int64_t hashed = (int64_t)t.hash_list.size();
            prefix_cache_mgr_.DecRefCount(t.hash_list.data(), hashed);
            page_mgr_.Free(t.page_list.data() + hashed, (int64_t)t.page_list.size() - hashed);
        } else {
            page_mgr_.Free(t.page_list.data(), (int64_t)t.page_list.size());
        }
        if (generator_config_.enable_penalty) batch_slots_mgr_.Free((uint64_t)in->batch_slots[row], 1);
        // (the reference adds to a by-value pointer here and so counts nothing -- SURVEY.md Q2; counted properly)
        worker_profiler_->req_counter.output_tokens_per_req += (uint64_t)t.gen_tokens_cnt;
        tid_data_map_.erase(it);
        ++worker_profiler_->finished_task_cnt;
    }
    CompactBatch(in);
}

// reference ReleaseResource, llm_generator.cc:368-385 (after a failed Execute)
void LLMGenerator::ReleaseResource() {
    for (TidData* t : tid_list_) {
        if (model_config_.cache_mode == 0) idx_mgr_.Free(t->cache_index, (uint64_t)t->total_len - 1);
        else page_mgr_.Free(t->page_list.data(), (int64_t)t->page_list.size());
        if (generator_config_.enable_penalty) batch_slots_mgr_.Free((uint64_t)t->slot_index, 1);
    }
    prefix_cache_mgr_.Reset();
    tid_list_.clear();
    tid_data_map_.clear();
    req_list_changed_ = true;
    FinishedTaskInfo info;
    while (finished_tasks_.Pop(&info)) {}
}

// reference DecodeAndSendTask, llm_generator.cc:58-112: one Response per request per step; text requests buffer up
// to three tokens while the piece decodes to U+FFFD
void LLMGenerator::SendTokens(const std::vector<TidGenToken>& tokens) {
    static const char kReplacement[] = "\xef\xbf\xbd";
    std::vector<Response> rsp(tokens.size());
    for (size_t i = 0; i < tokens.size(); ++i) {
        const TidGenToken& g = tokens[i];
        Response& r = rsp[i];
        r.id = g.tid;
        r.token = g.token;
        r.finish_flag = g.finish_flag;
        r.logprob = g.logprob;
        r.is_special = g.is_special;
        if (!g.is_token_in_out && tokenizer_) {
            int tok = g.token;
            tokenizer_->Decode(&tok, 1, &r.generated);
            if (r.generated == kReplacement) {
                std::vector<int>& buf = decode_buffer_[g.tid];
                buf.push_back(g.token);
                r.generated.clear();
                if (buf.size() == 3) {
                    tokenizer_->Decode(buf.data(), 3, &r.generated);
                    buf.clear();
                }
            }
            if (g.finish_flag != FinishFlag::NOT_FINISHED) decode_buffer_.erase(g.tid);
        }
    }
    conn_->Send(rsp);
}

// ------------------------------------------------------------------------------------------------ the step loop

// reference Generate, llm_generator.cc:574-786
void LLMGenerator::Generate() {
    ModelInput in;
    ModelOutput out;
    Admission adm;
    int running_batch = 0, prefill_batch = 0;
    int32_t cool_down = 0;
    uint64_t loop_step = 0;
    bool is_prefix_cache_hit = false;
    std::string error_msg;
    auto& counters = worker_profiler_->step_counter;

    tid_list_.clear();
    tid_data_map_.clear();
    req_list_changed_ = true;
    { FinishedTaskInfo drop; while (finished_tasks_.Pop(&drop)) {} }

    const std::function<bool(const LlmRequest&)> admit = [&](const LlmRequest& req) {
        return AdmitRequest(req, &adm, &cool_down, &is_prefix_cache_hit);
    };

    while (true) {
        is_prefix_cache_hit = false;
        const auto step_begin = std::chrono::high_resolution_clock::now();
        adm.total_tokens_per_step = running_batch;  // every running request contributes one decode token
        adm.running_batch = running_batch;
        adm.prefill_batch = 0;
        {
            utils::TimingGuard timing(&counters.current.prepare_cost);
            while (adm.running_batch < generator_config_.max_running_batch &&
                   adm.prefill_batch < generator_config_.max_prefill_batch && cool_down <= 0) {
                std::unique_ptr<LlmRequest> req(sched_.TryPopRequest(admit));
                if (!req) break;
                ++worker_profiler_->req_counter.waiting_cnt;
                worker_profiler_->req_counter.waiting_cost += (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(
                    std::chrono::high_resolution_clock::now() - req->enqueue_ts).count();
                if (!StartRequest(*req, adm, &in)) break;
                req_list_changed_ = true;
            }
            running_batch = (int)tid_list_.size();
            if (running_batch == 0) break;
            PackStep(req_list_changed_, &in);
            worker_profiler_->max_running_task = std::max<uint64_t>(running_batch, worker_profiler_->max_running_task);
            prefill_batch = adm.prefill_batch;
        }
        counters.global.prepare_cost += counters.current.prepare_cost;

        if (observer_) observer_(observer_arg_, loop_step, in, req_list_changed_, is_prefix_cache_hit);
        out.Clear();
        out.Resize(running_batch);
        error_msg.clear();
        const RetCode rc = llm_engine_.Execute(in, req_list_changed_, is_prefix_cache_hit, &out, &error_msg);
        if (rc != RC_SUCCESS) {
            LOG(ERROR) << "llm engine excute failed";
            for (TidData* t : tid_list_) conn_->NotifyFailure(t->tid, rc, error_msg);
            ReleaseResource();
            break;
        }
        req_list_changed_ = false;

        {
            utils::TimingGuard timing(&counters.current.post_process_cost);
            decoder_thread_pool_.Wait();  // the previous step's responses are out
            auto tokens = std::make_shared<std::vector<TidGenToken>>();
            tokens->reserve(running_batch);
            for (int row = 0; row < running_batch; ++row) {
                TidData* t = tid_list_[row];
                ++t->gen_tokens_cnt;
                const int tok = out.output_token[row];
                const int64_t fed = (int64_t)t->next_tokens->size();
                t->next_tokens = std::make_shared<std::vector<int>>(1, tok);
                if (t->steps == 0) {  // prefill row becomes a decode row
                    in.start_pos[row] += fed;
                    ++in.decoding_batches;
                } else {
                    ++in.start_pos[row];
                }
                t->start_pos += fed;
                ++t->steps;
                --t->rest_iters;
                FinishFlag flag = FinishFlag::NOT_FINISHED;
                const bool hit_stop = t->early_stopping &&
                    (generator_config_.stop_tokens.count(tok) || (t->stop_tokens && t->stop_tokens->count(tok)));
                if (t->rest_iters <= 0 || hit_stop) {
                    flag = t->rest_iters <= 0 ? FinishFlag::LENGTH : FinishFlag::EOS_TOKEN;
                    if (cool_down > 0) --cool_down;
                    finished_tasks_.Push(FinishedTaskInfo(t->tid, FinishedTaskInfo::FROM_WORKER));
                    req_list_changed_ = true;
                }
                tokens->push_back(TidGenToken{t->tid, tok, out.logprobs[row], flag, (uint64_t)t->steps, t->is_token_in_out,
                                              generator_config_.special_tokens.count(tok) > 0});
            }
            // detokenise + send overlaps the next step (llm_generator.cc:738-745)
            decoder_thread_pool_.RunAsync([this, tokens](uint32_t, uint32_t) { SendTokens(*tokens); });
            if (finished_tasks_.Size() > 0) DeleteTasks(&in);
        }
        counters.global.post_process_cost += counters.current.post_process_cost;

        counters.current.total_cost = (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(
            std::chrono::high_resolution_clock::now() - step_begin).count();
        counters.global.total_cost += counters.current.total_cost;
        worker_profiler_->pending_task_size = sched_.GetPendingSize();
        ++counters.global.step_cnt;
        ++loop_step;
        if (generator_config_.enable_profiling && (loop_step == 1 || loop_step % 100 == 0 || tid_list_.empty())) {
            worker_profiler_->running_task = running_batch;
            worker_profiler_->prefill_batch = prefill_batch;
            worker_profiler_->prefill_tokens = in.token_inputs.size() - (running_batch - prefill_batch);
            worker_profiler_->kv_max_blk = kv_cache_max_tokens_;
            worker_profiler_->kv_rest_blk = model_config_.cache_mode == 0
                ? (uint64_t)idx_mgr_.GetAvailableBlockNum()
                : (uint64_t)(page_mgr_.GetAvail() * model_config_.page_size);
            conn_->OnProfiling(worker_profiler_);
        }
    }
    decoder_thread_pool_.Wait();
}

}}  // namespace ppl::llm
