// Boundary proof for SURVEY.md 8(b) B2 (TEST INFRASTRUCTURE; built only where /root/reference exists).
//
// This driver is compiled together with the REFERENCE'S OWN sources, in place from /root/reference/src --
//   src/generator/llm_generator.cc, src/engine/llm_engine.cc, src/utils/utils.cc, src/common/profiler.cc
// -- against the ppl.nn / ppl.common surface of ppl.llm.serving_amd/src/compat (nothing of the reference is copied; the
// recipe is the `ref` target of ppl.llm.serving_amd/Makefile).  It runs the reference's LLMGenerator + LLMEngine over a
// fake ppl::nn::Runtime whose tensors record what the engine binds by index and copies per step
// (src/engine/llm_engine.h:124-147, src/engine/llm_engine.cc:29-116), and prints the same per-step JSON lines as
// tests/host/sched_trace.cc prints for the repo's generator.  tests/test_host_logic.py requires the two traces (and the
// oracle's, oracle/host_logic.py) to be identical: the repo's host side IS the reference's scheduling, and the compat
// surface is what the reference's engine needs from a backend.  It is not a parity pin of any arithmetic.
//
//   ref_sched_trace <scenario.json>
#include <condition_variable>
#include <cstdarg>
#include <fstream>
#include <iostream>
#include <mutex>
#include <sstream>
#include <thread>

#include "generator/llm_generator.h"                               // the reference's (include root /root/reference/src)
#include "ppl/nn/engines/llm_cuda/options.h"
#include "../../ppl.llm.serving_amd/src/utils/mini_json.h"          // the repo's scenario reader (std-only header)

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

class HostDevice final : public ppl::nn::DeviceContext {
public:
    const char* GetType() const override { return "cpu"; }
    RetCode Configure(uint32_t, ...) override { return RC_SUCCESS; }
};

// a tensor that keeps a host copy of whatever the engine copies into it
class FakeTensor final : public ppl::nn::Tensor {
public:
    explicit FakeTensor(const char* name, datatype_t dt) : name_(name) { shape_.SetDataType(dt); }
    const char* GetName() const override { return name_.c_str(); }
    ppl::nn::TensorShape* GetShape() const override { return const_cast<ppl::nn::TensorShape*>(&shape_); }
    ppl::nn::DeviceContext* GetDeviceContext() const override { return dev_; }
    void SetDeviceContext(ppl::nn::DeviceContext* d) override { dev_ = d; }
    void SetBufferPtr(void* p) override { ext_ = p; }
    void* GetBufferPtr() const override { return ext_ ? ext_ : (void*)data_.data(); }
    RetCode ReallocBuffer() override { data_.resize(shape_.CalcBytesIncludingPadding()); return RC_SUCCESS; }
    void FreeBuffer() override { ++frees_; }
    RetCode CopyFromHostAsync(const void* src) override {
        uint64_t bytes = shape_.CalcBytesIncludingPadding();
        if (bytes == 0) bytes = GetSizeOfDataType(shape_.GetDataType());  // the three scalars are never reshaped by the engine
        data_.assign((const char*)src, (const char*)src + bytes);
        ++copies_;
        return RC_SUCCESS;
    }
    RetCode CopyFromHost(const void* src) override { return CopyFromHostAsync(src); }
    RetCode CopyToHost(void* dst) const override { memcpy(dst, data_.data(), data_.size()); return RC_SUCCESS; }
    RetCode ConvertToHost(void* dst, const ppl::nn::TensorShape&) const override { return CopyToHost(dst); }

    std::vector<int64_t> AsI64() const { return std::vector<int64_t>((const int64_t*)data_.data(), (const int64_t*)(data_.data() + data_.size())); }
    int64_t Scalar() const { return data_.size() >= 8 ? *(const int64_t*)data_.data() : 0; }
    int copies_ = 0, frees_ = 0;

private:
    std::string name_;
    ppl::nn::TensorShape shape_;
    ppl::nn::DeviceContext* dev_ = nullptr;
    void* ext_ = nullptr;
    std::vector<char> data_;
};

struct TraceCtx;

// The fake model of tests/host/sched_trace.cc: next token of row b = (31 * last_input_token + 7 * (start_pos + seqlen) + 3) % vocab
class FakeRuntime final : public ppl::nn::Runtime {
public:
    FakeRuntime(int vocab, bool quant, TraceCtx* ctx) : vocab_(vocab), ctx_(ctx) {
        const char* names[11] = {"token_ids", "attn_mask", "seq_starts", "kv_starts", "cache_indices", "decoding_batches", "start_pos",
                                 "max_seq_len", "max_kv_len", "kv_cache", "kv_scale"};
        for (int i = 0; i < (quant ? 11 : 10); ++i)
            in_.emplace_back(new FakeTensor(names[i], i == 9 ? DATATYPE_INT8 : (i == 10 ? DATATYPE_FLOAT16 : DATATYPE_INT64)));
        logits_.reset(new FakeTensor("logits", DATATYPE_FLOAT32));
    }
    uint32_t GetInputCount() const override { return (uint32_t)in_.size(); }
    ppl::nn::Tensor* GetInputTensor(uint32_t i) const override { return i < in_.size() ? in_[i].get() : nullptr; }
    uint32_t GetOutputCount() const override { return 1; }
    ppl::nn::Tensor* GetOutputTensor(uint32_t i) const override { return i == 0 ? logits_.get() : nullptr; }
    uint32_t GetDeviceContextCount() const override { return 1; }
    ppl::nn::DeviceContext* GetDeviceContext(uint32_t) const override { return const_cast<HostDevice*>(&dev_); }
    RetCode Run() override;

    FakeTensor* in(int i) const { return in_[i].get(); }
    int fail_at_run_ = -1;

private:
    int vocab_;
    int runs_ = 0;
    TraceCtx* ctx_;
    HostDevice dev_;
    std::vector<std::unique_ptr<FakeTensor>> in_;
    std::unique_ptr<FakeTensor> logits_;
    std::vector<float> logit_store_;
};

class FakeEngine final : public ppl::nn::Engine {
public:
    const char* GetName() const override { return "fake"; }
    RetCode Configure(uint32_t option, ...) override {
        va_list ap;
        va_start(ap, option);
        if (option == ppl::nn::llm::cuda::ENGINE_CONF_CACHE_PREFILL) cache_prefill = va_arg(ap, int);
        va_end(ap);
        return RC_SUCCESS;
    }
    int cache_prefill = 0;
};

class FakePostProcessor final : public PostProcessor {
public:
    RetCode InitPostProcessorMem(int, int, bool) override { return RC_SUCCESS; }
    RetCode SampleTopKTopP(const float* logits, const float*, const int32_t*, const float*, int32_t batch, int32_t vocab, int32_t stride,
                           int32_t, float, bool req_list_changed, int32_t* out, float* lp, bool) override {
        last_changed = req_list_changed;
        for (int b = 0; b < batch; ++b) {
            int best = 0;
            for (int v = 1; v < vocab; ++v)
                if (logits[(size_t)b * stride + v] > logits[(size_t)b * stride + best]) best = v;
            out[b] = best;
            lp[b] = 0.f;
        }
        return RC_SUCCESS;
    }
    RetCode ApplyPenalty(const float*, const float*, const float*, const float*, const int64_t* slots, const int64_t*, const int64_t*,
                         const int64_t*, int32_t batch, int32_t, bool, float*) override {
        last_slots.assign(slots, slots + batch);
        return RC_SUCCESS;
    }
    bool last_changed = false;
    std::vector<int64_t> last_slots;
};

class RecordingConnection final : public Connection {
public:
    void OnProfiling(const std::shared_ptr<WorkerProfiler>&) override {}
    void OnTokenize(uint64_t, const std::vector<int>&) override {}
    void Send(const std::vector<Response>& rsps) override {
        std::lock_guard<std::mutex> g(mu_);
        for (const auto& r : rsps) {
            tokens_[r.id].push_back(r.token);
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
    bool WaitFor(size_t wanted, int ms) {
        std::unique_lock<std::mutex> lk(mu_);
        return cv_.wait_for(lk, std::chrono::milliseconds(ms), [&] { return done_ >= wanted; });
    }
    std::mutex mu_;
    std::condition_variable cv_;
    size_t done_ = 0;
    std::map<uint64_t, std::vector<int>> tokens_;
    std::map<uint64_t, int> finish_, failed_;
};

struct TraceCtx {
    LLMGenerator* gen = nullptr;
    FakeEngine* engine = nullptr;
    std::map<uint64_t, std::vector<uint64_t>> cancel_at;  // step -> ids
    int cache_mode = 0;
    uint64_t step = 0;
    int last_page_copies = 0;
    std::vector<int64_t> pages;  // the page table the "device" holds (refreshed only when the engine copies it)
    int64_t max_pages = 0;
    std::atomic<uint64_t> last_run_ms{0};
};

RetCode FakeRuntime::Run() {
    // what the engine handed over for this step = the ModelInput the reference's generator packed
    const std::vector<int64_t> tok = in(0)->AsI64(), seq = in(2)->AsI64(), kvs = in(3)->AsI64(), sp = in(6)->AsI64();
    const int64_t B = (int64_t)sp.size();
    bool changed = false;
    std::vector<int64_t> cache_indices;
    if (ctx_->cache_mode == 0) {
        cache_indices = in(4)->AsI64();
        changed = true;  // (mode 0 copies the indices every step; the flag is not observable here)
    } else if (in(4)->copies_ != ctx_->last_page_copies) {  // llm_engine.cc:67-71: only when the batch changed
        ctx_->last_page_copies = in(4)->copies_;
        ctx_->pages = in(4)->AsI64();
        ctx_->max_pages = in(4)->GetShape()->GetDimCount() == 2 ? in(4)->GetShape()->GetDim(1) : 0;
        changed = true;
    }
    std::ostringstream ss;
    ss << "{\"step\":" << ctx_->step << ",\"decoding_batches\":" << in(5)->Scalar() << ",\"max_seq_len\":" << in(7)->Scalar()
       << ",\"max_kv_len\":" << in(8)->Scalar() << ",\"max_pages\":" << (ctx_->cache_mode == 1 ? ctx_->max_pages : 0)
       << ",\"pages_uploaded\":" << (ctx_->cache_mode == 1 && changed ? 1 : 0) << ",\"prefix_hit\":" << ctx_->engine->cache_prefill
       << ",\"token_inputs\":" << Arr(tok) << ",\"seq_starts\":" << Arr(seq) << ",\"kv_starts\":" << Arr(kvs) << ",\"start_pos\":" << Arr(sp)
       << ",\"cache_indices\":" << Arr(cache_indices) << ",\"page_list\":" << Arr(ctx_->cache_mode == 1 ? ctx_->pages : std::vector<int64_t>()) << "}";
    std::cout << ss.str() << std::endl;
    auto it = ctx_->cancel_at.find(ctx_->step);
    if (it != ctx_->cancel_at.end())
        for (uint64_t id : it->second) ctx_->gen->ClearTask(id);
    ++ctx_->step;
    if (fail_at_run_ >= 0 && runs_++ == fail_at_run_) return RC_DEVICE_RUNTIME_ERROR;
    logit_store_.assign((size_t)B * vocab_, 0.f);
    for (int64_t b = 0; b < B; ++b) {
        const int64_t last = tok[seq[b + 1] - 1];
        const int64_t kv = sp[b] + (seq[b + 1] - seq[b]);
        logit_store_[(size_t)b * vocab_ + (31 * last + 7 * kv + 3) % vocab_] = 1.f;
    }
    logits_->GetShape()->Reshape({B, (int64_t)vocab_});
    logits_->SetBufferPtr(logit_store_.data());
    return RC_SUCCESS;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc != 2) {
        std::cerr << "usage: ref_sched_trace <scenario.json>\n";
        return 2;
    }
    std::ifstream ifs(argv[1]);
    std::stringstream buf;
    buf << ifs.rdbuf();
    ppl::llm::utils::JsonValue doc;
    if (!ppl::llm::utils::JsonParser(buf.str()).Parse(&doc)) {
        std::cerr << "bad scenario json\n";
        return 2;
    }
    const auto* m = doc.Find("model");
    const auto* g = doc.Find("generator");
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
    if (const auto* st = g->Find("stop_tokens"))
        for (const auto& v : st->arr) gc.stop_tokens.insert((int)v.AsInt());

    StaticThreadPool pool;
    pool.Init(1);
    TraceCtx tctx;
    tctx.cache_mode = mc.cache_mode;
    FakeRuntime rt(mc.vocab_size, mc.cache_quant_bit > 0, &tctx);
    rt.fail_at_run_ = (int)doc.GetInt("fail_at_run", -1);
    FakeEngine engine;
    tctx.engine = &engine;
    HostDevice host;
    FakePostProcessor pp;
    static char kv_dummy[64];
    Resource res;
    res.tensor_parallel_size = 1;
    res.kv_cache_max_tokens = (uint64_t)doc.GetInt("kv_cache_max_tokens", 4096);
    res.items.resize(1);
    res.items[0].kv_cache_mem = kv_dummy;
    res.items[0].kv_scale_mem = kv_dummy;
    res.items[0].runtime = &rt;
    res.items[0].host_device = &host;
    res.items[0].engine = &engine;
    res.post_processor = &pp;
    res.device_worker_pool_ = &pool;

    RecordingConnection conn;
    if (const auto* c = doc.Find("cancel"))
        for (const auto& e : c->arr) tctx.cancel_at[(uint64_t)e.GetInt("at_step", 0)].push_back((uint64_t)e.GetInt("id", 0));
    const size_t expect_done = (size_t)doc.GetInt("expect_done", -1);
    {
        LLMGenerator gen(res, gc, mc, &conn);
        tctx.gen = &gen;
        // all requests are queued BEFORE the generator thread starts: the admission order is then deterministic
        for (const auto& r : doc.Find("requests")->arr) {
            auto req = std::make_shared<Request>();
            req->id = (uint64_t)r.GetInt("id", 0);
            req->generation_length = (int32_t)r.GetInt("generation_length", 1);
            req->early_stopping = r.GetBool("early_stopping", true);
            req->token_ids = std::make_shared<std::vector<int>>();
            for (const auto& t : r.Find("tokens")->arr) req->token_ids->push_back((int)t.AsInt());
            if (const auto* st = r.Find("stop_tokens")) {
                req->stop_tokens = std::make_shared<std::unordered_set<int>>();
                for (const auto& v : st->arr) req->stop_tokens->insert((int)v.AsInt());
            }
            gen.Process(req);
        }
        if (gen.Init() != RC_SUCCESS) {
            std::cout << "{\"init_failed\":1}" << std::endl;
            return 0;
        }
        // the reference has no idle query: wait for the number of finished + failed requests the scenario expects
        conn.WaitFor(expect_done, 20000);
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
    }
    std::ostringstream ss;
    ss << "{\"responses\":{";
    bool first = true;
    for (auto& kv : conn.tokens_) {
        ss << (first ? "" : ",") << "\"" << kv.first << "\":{\"tokens\":" << Arr(kv.second) << ",\"finish\":"
           << (conn.finish_.count(kv.first) ? conn.finish_[kv.first] : 0) << "}";
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
