// Test driver for the host side of the hot path (TEST INFRASTRUCTURE: a fake backend, never shipped).
//
//   sched_trace <scenario.json>   runs LLMGenerator + LLMEngine against a scripted fake Runtime / PostProcessor and
//                                 prints one JSON line per step with the ModelInput the backend would receive, then one
//                                 line with everything the Connection received.
//   sched_trace --unit            prints known-answer values of HashCombine / PrefixCacheManager / IndexManager /
//                                 PageManager / params.json parsing / MPSC scheduler as one JSON object.
//
// The fake model: next token of row b = (31 * last_input_token + 7 * (start_pos + seqlen) + 3) % vocab  (restated in
// oracle/host_logic.py so that finish-by-stop-token paths are reproducible).
#include <condition_variable>
#include <fstream>
#include <iostream>
#include <mutex>
#include <sstream>
#include <thread>

#include "common/config.h"
#include "common/request.h"
#include "common/resource.h"
#include "generator/llm_generator.h"
#include "tokenizer/tokenizer_factory.h"
#include "utils/index_manager.h"
#include "utils/mini_json.h"
#include "utils/mpsc_request_scheduler.h"
#include "utils/prefix_cache_manager.h"
#include "utils/utils.h"

using namespace ppl::llm;
using namespace ppl::common;

namespace {

template <typename T>
std::string Arr(const std::vector<T>& v) {
    std::ostringstream ss;
    ss << "[";
    for (size_t i = 0; i < v.size(); ++i) ss << (i ? "," : "") << v[i];
    ss << "]";
    return ss.str();
}

class FakeRuntime final : public Runtime {
public:
    explicit FakeRuntime(int vocab) : vocab_(vocab) {}
    RetCode SetInputs(const StepInputs& in) override {
        B_ = in.batch;
        tok_.assign(in.token_inputs, in.token_inputs + in.num_tokens);
        seq_.assign(in.seq_starts, in.seq_starts + in.batch + 1);
        sp_.assign(in.start_pos, in.start_pos + in.batch);
        return RC_SUCCESS;
    }
    RetCode Run(bool) override {
        if (fail_at_run_ >= 0 && runs_++ == fail_at_run_) return RC_DEVICE_RUNTIME_ERROR;
        logits_.assign((size_t)B_ * vocab_, 0.f);
        for (int64_t b = 0; b < B_; ++b) {
            const int64_t last = tok_[seq_[b + 1] - 1];
            const int64_t kv = sp_[b] + (seq_[b + 1] - seq_[b]);
            int64_t t = (31 * last + 7 * kv + 3) % vocab_;
            auto it = chain_.find(last);            // scripted continuation (detokeniser tests): token -> next token
            if (it != chain_.end()) t = it->second;
            logits_[(size_t)b * vocab_ + t] = 1.f;
        }
        return RC_SUCCESS;
    }
    float* GetLogits(int64_t* stride) override {
        *stride = vocab_;
        return logits_.data();
    }
    int fail_at_run_ = -1;
    std::map<int64_t, int64_t> chain_;

private:
    int vocab_;
    int runs_ = 0;
    int64_t B_ = 0;
    std::vector<int64_t> tok_, seq_, sp_;
    std::vector<float> logits_;
};

class FakePostProcessor final : public PostProcessor {
public:
    RetCode InitPostProcessorMem(int, int, bool) override { return RC_SUCCESS; }
    RetCode SampleTopKTopP(const float* logits, const float*, const int32_t*, const float*, int32_t batch, int32_t vocab,
                           int32_t stride, int32_t, float, bool, int32_t* out, float* lp, bool) override {
        for (int b = 0; b < batch; ++b) {
            int best = 0;
            for (int v = 1; v < vocab; ++v)
                if (logits[(size_t)b * stride + v] > logits[(size_t)b * stride + best]) best = v;
            out[b] = best;
            lp[b] = 0.f;
        }
        return RC_SUCCESS;
    }
    RetCode ApplyPenalty(const float*, const float*, const float*, const float*, const int64_t*, const int64_t*, const int64_t*,
                         const int64_t*, int32_t, int32_t, bool, float*) override {
        return RC_SUCCESS;
    }
};

class RecordingConnection final : public Connection {
public:
    void OnProfiling(const std::shared_ptr<WorkerProfiler>&) override {}
    void OnTokenize(uint64_t, const std::vector<int>&) override {}
    void Send(const std::vector<Response>& rsps) override {
        std::lock_guard<std::mutex> g(mu_);
        for (const auto& r : rsps) {
            tokens_[r.id].push_back(r.token);
            texts_[r.id].push_back(r.generated);
            if (r.finish_flag != FinishFlag::NOT_FINISHED) {
                finish_[r.id] = (int)r.finish_flag;
                ++done_;
            }
        }
        cv_.notify_all();
    }
    void NotifyFailure(uint64_t id, RetCode rc, const std::string&) override {
        std::lock_guard<std::mutex> g(mu_);
        failed_[id] = (int)rc;
        ++done_;
        cv_.notify_all();
    }
    void Wait(size_t wanted) {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return done_ >= wanted; });
    }
    std::mutex mu_;
    std::condition_variable cv_;
    size_t done_ = 0;
    std::map<uint64_t, std::vector<int>> tokens_;
    std::map<uint64_t, std::vector<std::string>> texts_;   // Response::generated per response (text requests)
    std::map<uint64_t, int> finish_, failed_;
};

struct TraceCtx {
    LLMGenerator* gen = nullptr;
    std::map<uint64_t, std::vector<uint64_t>> cancel_at;  // step -> ids
    int cache_mode = 0;
};

void Observe(void* arg, uint64_t step, const ModelInput& in, bool changed, bool hit) {
    auto* t = static_cast<TraceCtx*>(arg);
    std::ostringstream ss;
    ss << "{\"step\":" << step << ",\"decoding_batches\":" << in.decoding_batches << ",\"max_seq_len\":" << in.max_seq_len
       << ",\"max_kv_len\":" << in.max_kv_len << ",\"max_pages\":" << in.max_pages << ",\"req_list_changed\":" << (changed ? 1 : 0)
       << ",\"prefix_hit\":" << (hit ? 1 : 0) << ",\"token_inputs\":" << Arr(in.token_inputs) << ",\"seq_starts\":" << Arr(in.seq_starts)
       << ",\"kv_starts\":" << Arr(in.kv_starts) << ",\"start_pos\":" << Arr(in.start_pos)
       << ",\"cache_indices\":" << Arr(in.cache_indices) << ",\"page_list\":" << Arr(in.page_list)
       << ",\"batch_slots\":" << Arr(in.batch_slots) << "}";
    std::cout << ss.str() << std::endl;
    auto it = t->cancel_at.find(step);
    if (it != t->cancel_at.end())
        for (uint64_t id : it->second) t->gen->ClearTask(id);
}

int RunScenario(const char* path) {
    std::ifstream ifs(path);
    std::stringstream buf;
    buf << ifs.rdbuf();
    utils::JsonValue doc;
    if (!utils::JsonParser(buf.str()).Parse(&doc)) {
        std::cerr << "bad scenario json\n";
        return 2;
    }
    const utils::JsonValue* m = doc.Find("model");
    const utils::JsonValue* g = doc.Find("generator");
    ModelConfig mc;
    mc.hidden_dim = 64; mc.intermediate_dim = 64; mc.num_layers = 1; mc.num_heads = 2; mc.num_kv_heads = 2;
    mc.vocab_size = (int32_t)m->GetInt("vocab_size", 1000);
    mc.cache_quant_bit = (int32_t)m->GetInt("cache_quant_bit", 8);
    mc.cache_quant_group = (int32_t)m->GetInt("cache_quant_group", 8);
    mc.cache_layout = (int32_t)m->GetInt("cache_layout", 3);
    mc.cache_mode = (int32_t)m->GetInt("cache_mode", 0);
    mc.page_size = (int32_t)m->GetInt("page_size", 0);
    GeneratorConfig gc;
    gc.top_k = 1;
    gc.max_running_batch = (int32_t)g->GetInt("max_running_batch", 1024);
    gc.max_input_tokens_per_request = (int32_t)g->GetInt("max_input_tokens_per_request", 4096);
    gc.max_output_tokens_per_request = (int32_t)g->GetInt("max_output_tokens_per_request", 4096);
    gc.max_total_tokens_per_request = (int32_t)g->GetInt("max_total_tokens_per_request", 8192);
    gc.max_tokens_per_step = (int32_t)g->GetInt("max_tokens_per_step", 8192);
    gc.max_cooldown_request = (int)g->GetInt("max_cooldown_request", 2);
    gc.enable_prefix_cache = g->GetBool("enable_prefix_cache", false);
    gc.max_prefill_batch = (int32_t)g->GetInt("max_prefill_batch", 64);
    gc.enable_penalty = g->GetBool("enable_penalty", false);
    if (gc.enable_prefix_cache) gc.max_prefill_batch = 1;  // tools/offline_inference.cc:97-99
    if (const utils::JsonValue* st = g->Find("stop_tokens"))
        for (const auto& v : st->arr) gc.stop_tokens.insert((int)v.AsInt());

    StaticThreadPool pool;
    pool.Init(1);
    FakeRuntime rt(mc.vocab_size);
    rt.fail_at_run_ = (int)doc.GetInt("fail_at_run", -1);
    if (const utils::JsonValue* ch = doc.Find("chain"))
        for (const auto& e : ch->arr) rt.chain_[e.GetInt("from", -1)] = e.GetInt("to", 0);
    std::unique_ptr<Tokenizer> tokenizer;
    const std::string tok_path = doc.GetString("tokenizer", "");
    if (!tok_path.empty()) {
        tokenizer.reset(TokenizerFactory::Create("llama", "sentencepiece", tok_path, ""));
        if (!tokenizer) { std::cerr << "cannot load tokenizer " << tok_path << "\n"; return 2; }
    }
    FakePostProcessor pp;
    Resource res;
    res.tensor_parallel_size = 1;
    res.kv_cache_max_tokens = (uint64_t)doc.GetInt("kv_cache_max_tokens", 4096);
    res.items.resize(1);
    res.items[0].runtime = &rt;
    res.post_processor = &pp;
    res.device_worker_pool_ = &pool;
    res.tokenizer = tokenizer.get();

    RecordingConnection conn;
    TraceCtx tctx;
    tctx.cache_mode = mc.cache_mode;
    if (const utils::JsonValue* c = doc.Find("cancel"))
        for (const auto& e : c->arr) tctx.cancel_at[(uint64_t)e.GetInt("at_step", 0)].push_back((uint64_t)e.GetInt("id", 0));
    size_t n_req = 0;
    {
        LLMGenerator gen(res, gc, mc, &conn);
        tctx.gen = &gen;
        gen.SetStepObserver(Observe, &tctx);
        // all requests are queued BEFORE the generator thread starts: the admission order is then deterministic
        for (const auto& r : doc.Find("requests")->arr) {
            auto req = std::make_shared<Request>();
            req->id = (uint64_t)r.GetInt("id", 0);
            req->generation_length = (int32_t)r.GetInt("generation_length", 1);
            req->early_stopping = r.GetBool("early_stopping", true);
            if (const utils::JsonValue* toks = r.Find("tokens")) {
                req->token_ids = std::make_shared<std::vector<int>>();
                for (const auto& t : toks->arr) req->token_ids->push_back((int)t.AsInt());
            } else {
                req->prompt = r.GetString("prompt", "");     // text request: tokenised by the generator (Process)
            }
            if (const utils::JsonValue* st = r.Find("stop_tokens")) {
                req->stop_tokens = std::make_shared<std::unordered_set<int>>();
                for (const auto& v : st->arr) req->stop_tokens->insert((int)v.AsInt());
            }
            gen.Process(req);
            ++n_req;
        }
        if (gen.Init() != RC_SUCCESS) {
            std::cout << "{\"init_failed\":1}" << std::endl;
            return 0;
        }
        (void)n_req;
        while (!gen.IsIdle()) std::this_thread::sleep_for(std::chrono::milliseconds(1));  // cancelled requests never finish
        std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
    std::ostringstream ss;
    ss << "{\"responses\":{";
    bool first = true;
    for (auto& kv : conn.tokens_) {
        ss << (first ? "" : ",") << "\"" << kv.first << "\":{\"tokens\":" << Arr(kv.second) << ",\"finish\":"
           << (conn.finish_.count(kv.first) ? conn.finish_[kv.first] : 0) << "}";
        first = false;
    }
    ss << "},\"texts_hex\":{";
    first = true;
    for (auto& kv : conn.texts_) {
        ss << (first ? "" : ",") << "\"" << kv.first << "\":[";
        for (size_t i = 0; i < kv.second.size(); ++i) {
            ss << (i ? "," : "") << "\"";
            for (unsigned char c : kv.second[i]) { static const char* d = "0123456789abcdef"; ss << d[c >> 4] << d[c & 15]; }
            ss << "\"";
        }
        ss << "]";
        first = false;
    }
    ss << "},\"failed\":{";
    first = true;
    for (auto& kv : conn.failed_) {
        ss << (first ? "" : ",") << "\"" << kv.first << "\":" << kv.second;
        first = false;
    }
    ss << "}}";
    std::cout << ss.str() << std::endl;
    return 0;
}

struct Node final : public MPSCQueue::Node {
    int v = 0;
};

int RunUnit() {
    std::ostringstream ss;
    ss << "{";
    {  // HashCombine known answers (values captured from the reference's own function: SURVEY.md 8(c) C6)
        const int32_t a[] = {1, 2, 3, 4, 5};
        const int32_t b[] = {-1, -2, 2147483647, 0};
        int32_t page[16];
        for (int i = 0; i < 16; ++i) page[i] = 1000 + i;
        const uint64_t h1 = utils::HashCombine(0, page, 16);
        ss << "\"hash_a\":\"" << utils::HashCombine(0, a, 5) << "\",\"hash_b\":\"" << utils::HashCombine(7, b, 4) << "\",\"hash_p1\":\""
           << h1 << "\",\"hash_p2\":\"" << utils::HashCombine(h1, page, 16) << "\"";
    }
    {  // PrefixCacheManager: insert 4, release all, evict 2 -> oldest released first
        utils::PrefixCacheManager p;
        const uint64_t h[] = {0, 1, 2, 3};
        for (int i = 0; i < 4; ++i) p.Insert(h[i], 11 + i);
        p.DecRefCount(h, 4);
        std::vector<int64_t> ev;
        p.Evict(2, &ev);
        ss << ",\"prefix_evicted\":" << Arr(ev) << ",\"prefix_size\":" << p.Size() << ",\"prefix_find2\":" << p.Find(2)
           << ",\"prefix_find0\":" << p.Find(0);
        // re-reference keeps a page out of the LRU
        const uint64_t two = 2;
        p.IncRefCount(&two, 1);
        ev.clear();
        p.Evict(5, &ev);
        ss << ",\"prefix_evicted2\":" << Arr(ev) << ",\"prefix_size2\":" << p.Size();
    }
    {  // IndexManager first fit + coalescing; PageManager order
        utils::IndexManager im;
        im.Init(100);
        std::vector<int64_t> r;
        r.push_back(im.Alloc(30));
        r.push_back(im.Alloc(30));
        r.push_back(im.Alloc(30));
        r.push_back(im.Alloc(30));  // fails
        im.Free(30, 30);
        r.push_back(im.Alloc(10));
        r.push_back(im.Alloc(25));  // does not fit the 20-slot hole nor the 10-slot tail
        im.Free(0, 30);
        ss << ",\"index_allocs\":" << Arr(r) << ",\"index_avail\":" << im.GetAvailableBlockNum();
        PageManager pm;
        pm.Init(35, 8);  // 4 pages
        std::vector<int64_t> pages;
        pm.Alloc(3, &pages);
        const RetCode rc = pm.Alloc(2, &pages);
        const int64_t two[] = {pages[1]};
        pm.Free(two, 1);
        pm.Alloc(2, &pages);
        ss << ",\"pages\":" << Arr(pages) << ",\"pages_rc\":" << rc << ",\"pages_avail\":" << pm.GetAvail();
    }
    {  // params.json
        ModelConfig mc;
        const bool ok = ParseModelConfigFromString(
            "{\"num_heads\": 32, \"num_layers\": 32, \"hidden_dim\": 4096, \"intermediate_dim\": 11008, \"vocab_size\": 32000,"
            " \"cache_quant_bit\": 8, \"cache_quant_group\": 8, \"cache_layout\": 3, \"cache_mode\": 1, \"page_size\": 16,"
            " \"dynamic_batching\": true, \"auto_causal\": true, \"weight_quant_bit\": 8}", &mc);
        ModelConfig bad;
        const bool ok2 = ParseModelConfigFromString("{\"num_heads\": 32}", &bad);
        const bool ok3 = ParseModelConfigFromString(
            "{\"num_heads\": 8, \"num_kv_heads\": 2, \"num_layers\": 1, \"hidden_dim\": 64, \"intermediate_dim\": 64, \"vocab_size\": 10,"
            " \"cache_quant_bit\": 0, \"cache_quant_group\": 1, \"cache_layout\": 0, \"cache_mode\": 1, \"dynamic_batching\": true,"
            " \"auto_causal\": true}", &bad);  // page_size missing with cache_mode 1
        ss << ",\"cfg_ok\":" << ok << ",\"cfg_kv_heads\":" << mc.num_kv_heads << ",\"cfg_page\":" << mc.page_size << ",\"cfg_wq\":"
           << mc.weight_quant_bit << ",\"cfg_missing\":" << ok2 << ",\"cfg_nopage\":" << ok3;
    }
    {  // MPSC scheduler: stash semantics + multi-producer push
        utils::MPSCRequestScheduler<Node> s;
        std::vector<std::thread> th;
        for (int t = 0; t < 4; ++t)
            th.emplace_back([&s, t] {
                for (int i = 0; i < 250; ++i) {
                    auto* n = new Node();
                    n->v = t * 1000 + i;
                    s.PushRequest(n);
                }
            });
        for (auto& t : th) t.join();
        int popped = 0, rejected_first = -1, after = -2;
        std::vector<int> last(4, -1);
        bool fifo = true;
        // reject the head once: the same request must come back first
        Node* n = s.TryPopRequest([&](const Node& x) { rejected_first = x.v; return false; });
        (void)n;
        n = s.TryPopRequest([&](const Node& x) { after = x.v; return true; });
        if (n) { ++popped; last[n->v / 1000] = n->v % 1000; delete n; }
        while ((n = s.TryPopRequest([](const Node&) { return true; }))) {
            if (n->v % 1000 <= last[n->v / 1000]) fifo = false;
            last[n->v / 1000] = n->v % 1000;
            ++popped;
            delete n;
        }
        ss << ",\"mpsc_popped\":" << popped << ",\"mpsc_stash_same\":" << (rejected_first == after) << ",\"mpsc_fifo\":" << fifo
           << ",\"mpsc_pending\":" << s.GetPendingSize();
    }
    {
        std::set<int> toks;
        utils::ParseTokens("2,,13,7", &toks);
        ss << ",\"parse_tokens\":" << Arr(std::vector<int>(toks.begin(), toks.end()));
    }
    ss << "}";
    std::cout << ss.str() << std::endl;
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc == 2 && std::string(argv[1]) == "--unit") return RunUnit();
    if (argc == 2) return RunScenario(argv[1]);
    std::cerr << "usage: sched_trace <scenario.json> | --unit\n";
    return 2;
}
