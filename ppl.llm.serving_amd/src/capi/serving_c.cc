#include "capi/serving_c.h"

#include <chrono>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>

#include "backends/hip/resource_manager.h"
#include "common/config.h"
#include "common/request.h"
#include "common/resource.h"
#include "generator/llm_generator.h"
#include "tokenizer/tokenizer_factory.h"

using namespace ppl::llm;
using ppl::common::RetCode;

namespace {

class QueueConnection final : public Connection {
public:
    void OnProfiling(const std::shared_ptr<WorkerProfiler>&) override {}
    void OnTokenize(uint64_t, const std::vector<int>&) override {}
    void Send(const std::vector<Response>& batch) override {
        std::lock_guard<std::mutex> g(mu_);
        for (const auto& r : batch) {
            pplsrv_response o{};
            o.id = r.id;
            o.token = r.token;
            o.logprob = r.logprob;
            o.is_special = r.is_special ? 1 : 0;
            o.status = r.finish_flag == FinishFlag::NOT_FINISHED ? PPLSRV_PROCESSING : PPLSRV_FINISHED;
            o.finish_reason = r.finish_flag == FinishFlag::EOS_TOKEN ? PPLSRV_REASON_EOS
                            : r.finish_flag == FinishFlag::STOP_SEQUENCE ? PPLSRV_REASON_STOP : PPLSRV_REASON_LENGTH;
            o.text_off = -1;
            q_.push_back(o);
            text_.push_back(r.generated);
        }
        cv_.notify_all();
    }
    void NotifyFailure(uint64_t id, RetCode, const std::string&) override {
        std::lock_guard<std::mutex> g(mu_);
        pplsrv_response o{};
        o.id = id;
        o.status = PPLSRV_FAILED;
        o.text_off = -1;
        q_.push_back(o);
        text_.emplace_back();
        cv_.notify_all();
    }
    int Poll(pplsrv_response* out, int max, int timeout_ms, char* text_buf, int64_t text_cap) {
        std::unique_lock<std::mutex> lk(mu_);
        if (q_.empty() && timeout_ms > 0) cv_.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&] { return !q_.empty(); });
        int n = 0;
        int64_t used = 0;
        while (n < max && !q_.empty()) {
            const std::string& t = text_.front();
            // a response whose text does not fit any more stays queued for the next call (unless it could never fit)
            if (text_buf && !t.empty() && used + (int64_t)t.size() > text_cap && n > 0) break;
            out[n] = q_.front();
            if (text_buf && !t.empty() && used + (int64_t)t.size() <= text_cap) {
                memcpy(text_buf + used, t.data(), t.size());
                out[n].text_off = used;
                out[n].text_len = (int32_t)t.size();
                used += (int64_t)t.size();
            }
            ++n;
            q_.pop_front();
            text_.pop_front();
        }
        return n;
    }

private:
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<pplsrv_response> q_;
    std::deque<std::string> text_;  // Response::generated of q_[i]
};

}  // namespace

struct pplsrv {
    // destruction order matters: generator first, then the resources it points into (offline_inference.cc:414)
    hip::HipResourceManager resource_manager;
    Resource resource;
    QueueConnection conn;
    std::unique_ptr<Tokenizer> tokenizer;
    std::unique_ptr<LLMGenerator> generator;
};

extern "C" {

int pplsrv_create(const pplsrv_config* cfg, pplsrv** out) {
    if (!cfg || !out || !cfg->model_param_path) return -(int)ppl::common::RC_INVALID_VALUE;
    ResourceConfig rc;
    GeneratorConfig gc;
    ModelConfig mc;
    rc.model_type = "llama";
    rc.model_format = "pplhip";
    rc.model_dir = cfg->model_dir ? cfg->model_dir : "";
    rc.model_param_path = cfg->model_param_path;
    rc.tensor_parallel_size = cfg->tensor_parallel_size > 0 ? cfg->tensor_parallel_size : 1;
    rc.max_tokens_scale = cfg->max_tokens_scale > 0 ? cfg->max_tokens_scale : 0.94f;
    rc.max_running_batch = cfg->max_running_batch > 0 ? cfg->max_running_batch : 1024;
    rc.max_tokens_per_step = cfg->max_tokens_per_step > 0 ? cfg->max_tokens_per_step : 8192;
    rc.enable_penalty = cfg->enable_penalty != 0;
    rc.synthetic_weights = cfg->synthetic_weights != 0;
    rc.synthetic_seed = cfg->synthetic_seed;
    rc.kv_cache_max_tokens_override = cfg->kv_cache_max_tokens;
    rc.engine_config.configure_decoding_attn_split_k = cfg->decoding_attn_split_k > 0 ? cfg->decoding_attn_split_k - 1 : 1;
    rc.engine_config.specify_decoding_attn_tpb = cfg->decoding_attn_tpb;
    if (cfg->quant_method && cfg->quant_method[0]) rc.engine_config.quant_method = cfg->quant_method;
    gc.top_p = cfg->top_p;
    gc.top_k = cfg->top_k > 0 ? cfg->top_k : 1;
    gc.enable_penalty = rc.enable_penalty;
    gc.max_running_batch = rc.max_running_batch;
    gc.max_tokens_per_step = rc.max_tokens_per_step;
    gc.max_input_tokens_per_request = cfg->max_input_tokens_per_request > 0 ? cfg->max_input_tokens_per_request : 4096;
    gc.max_output_tokens_per_request = cfg->max_output_tokens_per_request > 0 ? cfg->max_output_tokens_per_request : 4096;
    gc.max_total_tokens_per_request = cfg->max_total_tokens_per_request > 0 ? cfg->max_total_tokens_per_request : 8192;
    gc.max_cooldown_request = cfg->max_cooldown_request > 0 ? cfg->max_cooldown_request : 2;
    gc.enable_prefix_cache = cfg->enable_prefix_cache != 0;
    gc.max_prefill_batch = gc.enable_prefix_cache ? 1 : (cfg->max_prefill_batch > 0 ? cfg->max_prefill_batch : 64);
    for (int i = 0; i < cfg->n_stop_tokens; ++i) gc.stop_tokens.insert(cfg->stop_tokens[i]);
    if (!ParseModelConfig(rc.model_param_path, &mc)) return -(int)ppl::common::RC_INVALID_VALUE;

    std::unique_ptr<pplsrv> s(new pplsrv());
    RetCode st = s->resource_manager.Init(mc, rc);
    if (st != ppl::common::RC_SUCCESS) return -(int)st;
    s->resource_manager.FillResource(&s->resource);
    if (cfg->tokenizer_path && cfg->tokenizer_path[0]) {  // tools/llm_server.cc: the tokenizer belongs to the Resource
        s->tokenizer.reset(TokenizerFactory::Create(cfg->model_type && cfg->model_type[0] ? cfg->model_type : "llama",
                                                    cfg->tokenizer_type && cfg->tokenizer_type[0] ? cfg->tokenizer_type : "sentencepiece",
                                                    cfg->tokenizer_path, ""));
        if (!s->tokenizer) return -(int)ppl::common::RC_INVALID_VALUE;
        s->resource.tokenizer = s->tokenizer.get();
    }
    s->generator.reset(new LLMGenerator(s->resource, gc, mc, &s->conn));
    st = s->generator->Init();
    if (st != ppl::common::RC_SUCCESS) return -(int)st;
    *out = s.release();
    return 0;
}

int pplsrv_submit(pplsrv* s, const pplsrv_request* reqs, int32_t n) {
    if (!s || (n > 0 && !reqs)) return -(int)ppl::common::RC_INVALID_VALUE;
    for (int i = 0; i < n; ++i) {
        const pplsrv_request& q = reqs[i];
        auto r = std::make_shared<Request>();
        r->id = q.id;
        r->temperature = q.temperature;
        r->top_p = q.top_p;
        r->top_k = q.top_k;
        r->repetition_penalty = q.repetition_penalty;
        r->presence_penalty = q.presence_penalty;
        r->frequency_penalty = q.frequency_penalty;
        r->generation_length = q.generation_length;
        r->early_stopping = q.early_stopping != 0;
        if (!q.tokens && q.prompt) {
            // text request (grpc_server.cc:218-252): LLMGenerator::Process tokenises it and adds the EOS id to its stop tokens
            if (!s->resource.tokenizer) {
                s->conn.NotifyFailure(q.id, ppl::common::RC_INVALID_VALUE, "no tokenizer configured");
                continue;
            }
            r->prompt.assign(q.prompt, q.n_prompt > 0 ? (size_t)q.n_prompt : 0);
        } else {
            r->is_token_in_out = true;
            r->token_ids = std::make_shared<std::vector<int>>(q.tokens, q.tokens + (q.n_tokens > 0 ? q.n_tokens : 0));
            r->stop_tokens = std::make_shared<std::unordered_set<int>>();  // grpc_server.cc:225 builds an empty set as well
        }
        s->generator->Process(r);
    }
    return 0;
}

int pplsrv_poll(pplsrv* s, pplsrv_response* out, int32_t max, int32_t timeout_ms) {
    if (!s || !out || max <= 0) return 0;
    return s->conn.Poll(out, max, timeout_ms, nullptr, 0);
}

int pplsrv_poll_text(pplsrv* s, pplsrv_response* out, int32_t max, int32_t timeout_ms, char* text_buf, int64_t text_buf_bytes) {
    if (!s || !out || max <= 0) return 0;
    return s->conn.Poll(out, max, timeout_ms, text_buf, text_buf ? text_buf_bytes : 0);
}

int pplsrv_cancel(pplsrv* s, uint64_t id) {
    if (!s) return -(int)ppl::common::RC_INVALID_VALUE;
    s->generator->ClearTask(id);
    return 0;
}

uint64_t pplsrv_kv_cache_max_tokens(pplsrv* s) { return s ? s->resource.kv_cache_max_tokens : 0; }

void pplsrv_destroy(pplsrv* s) {
    if (!s) return;
    s->generator.reset();
    delete s;
}

}  // extern "C"
