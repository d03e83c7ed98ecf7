// Shared by the offline tools: the reference's flag set (tools/offline_inference.cc:40-90), config structs filled from
// it (:194-232) and an in-process Connection that records per-request timing (LocalConnection, :234-301).
#pragma once
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <unordered_map>

#include "backends/hip/resource_manager.h"
#include "common/config.h"
#include "common/request.h"
#include "generator/llm_generator.h"
#include "simple_args.h"
#include "utils/utils.h"

namespace tools {

inline void DefineCommonFlags(Args* a) {
    a->Def("--help", "false", "show this help", true);
    a->Def("--model-type", "llama", "");
    a->Def("--model-format", "pplhip", "weights.pplhip slices (a ppl.pmx export, the reference's onnx / pmx, is converted by tools/import_pmx_onnx.py)");
    a->Def("--model-dir", "", "directory holding model_slice_<rank>/weights.pplhip");
    a->Def("--model-param-path", "", "params.json");
    a->Def("--tensor-parallel-size", "1", "");
    a->Def("--enable-penalty", "false", "whether enable penalty", true);
    a->Def("--max-tokens-scale", "0.94", "");
    a->Def("--top-p", "0.0", "");
    a->Def("--top-k", "1", "");
    a->Def("--max-input-tokens-per-request", "4096", "");
    a->Def("--max-output-tokens-per-request", "4096", "");
    a->Def("--max-total-tokens-per-request", "8192", "");
    a->Def("--max-running-batch", "1024", "");
    a->Def("--max-tokens-per-step", "8192", "");
    a->Def("--max-cooldown-request", "2", "when gpu mem is full, wait for this number of tasks to complete");
    a->Def("--quant-method", "none", "");
    a->Def("--cublas-layout-hint", "default", "accepted for CLI compatibility; ignored by the hip backend");
    a->Def("--disable-decoding-shm-mha", "false", "accepted; ignored", true);
    a->Def("--disable-decoding-inf-mha", "false", "accepted; ignored", true);
    a->Def("--disable-decoding-inf-gqa", "false", "accepted; ignored", true);
    a->Def("--configure-decoding-attn-split-k", "1", "always-on(2)/heuristic(1)/off(0)");
    a->Def("--specify-decoding-attn-tpb", "0", "512/256/heuristic(0)");
    a->Def("--disable-graph-fusion", "false", "accepted; ignored", true);
    a->Def("--stop-tokens", "", "stop tokens list");
    a->Def("--special_tokens", "", "special tokens");
    a->Def("--tokenizer-path", "", "tokenizer.model (sentencepiece); empty: token-in/token-out only");
    a->Def("--tokenizer-type", "sentencepiece", "sentencepiece (the huggingface json tokenizer is not provided)");
    a->Def("--tokenizer-config-path", "", "unused by the sentencepiece tokenizer");
    a->Def("--enable-prefix-cache", "false", "is enable prefix cache", true);
    a->Def("--max-prefill-batch", "64", "max prefill batches per step");
    a->Def("--enable-profiling", "false", "print profiling message", true);
    // flags of the reference's tools that only its server reads: accepted so that a command line carries over unchanged
    a->Def("--host", "127.0.0.1", "accepted; the offline tools open no socket");
    a->Def("--port", "10086", "accepted; ignored");
    a->Def("--monitor-port", "23333", "accepted; ignored");
    a->Def("--control-port", "12345", "accepted; ignored");
    a->Def("--version", "false", "accepted; ignored", true);
    // additions of this build
    a->Def("--synthetic-weights", "false", "fill the model slices with the synthetic generator instead of loading", true);
    a->Def("--synthetic-seed", "1234", "");
    a->Def("--synthetic-decisive-head", "0", "with --synthetic-weights: lm_head row v = embedding row v - N, i.e. token t is answered by t + N with a wide margin (answers of two runs can then be compared token for token); 0: off");
    a->Def("--kv-cache-max-tokens", "0", "pin the KV slab size in tokens (0: max-tokens-scale x free memory)");
    a->Def("--seed", "1234", "workload seed");
}

inline bool FillConfigs(const Args& a, ppl::llm::ResourceConfig* rc, ppl::llm::GeneratorConfig* gc, ppl::llm::ModelConfig* mc) {
    rc->model_type = a.Str("--model-type");
    rc->model_format = a.Str("--model-format");
    rc->model_dir = a.Str("--model-dir");
    rc->model_param_path = a.Str("--model-param-path");
    rc->tensor_parallel_size = a.Int("--tensor-parallel-size");
    rc->max_tokens_scale = (float)a.Num("--max-tokens-scale");
    rc->max_running_batch = a.Int("--max-running-batch");
    rc->max_tokens_per_step = a.Int("--max-tokens-per-step");
    rc->enable_penalty = a.Bool("--enable-penalty");
    rc->synthetic_weights = a.Bool("--synthetic-weights");
    rc->synthetic_seed = (uint64_t)a.I64("--synthetic-seed");
    rc->synthetic_decisive_head = a.I64("--synthetic-decisive-head");
    rc->kv_cache_max_tokens_override = (uint64_t)a.I64("--kv-cache-max-tokens");
    rc->engine_config.cublas_layout_hint = a.Str("--cublas-layout-hint");
    rc->engine_config.disable_graph_fusion = a.Bool("--disable-graph-fusion");
    rc->engine_config.disable_decoding_shm_mha = a.Bool("--disable-decoding-shm-mha");
    rc->engine_config.disable_decoding_inf_mha = a.Bool("--disable-decoding-inf-mha");
    rc->engine_config.disable_decoding_inf_gqa = a.Bool("--disable-decoding-inf-gqa");
    rc->engine_config.configure_decoding_attn_split_k = a.Int("--configure-decoding-attn-split-k");
    rc->engine_config.specify_decoding_attn_tpb = a.Int("--specify-decoding-attn-tpb");
    rc->engine_config.quant_method = a.Str("--quant-method");

    gc->top_p = (float)a.Num("--top-p");
    gc->top_k = a.Int("--top-k");
    gc->enable_penalty = a.Bool("--enable-penalty");
    gc->max_running_batch = a.Int("--max-running-batch");
    gc->max_input_tokens_per_request = a.Int("--max-input-tokens-per-request");
    gc->max_output_tokens_per_request = a.Int("--max-output-tokens-per-request");
    gc->max_total_tokens_per_request = a.Int("--max-total-tokens-per-request");
    gc->max_tokens_per_step = a.Int("--max-tokens-per-step");
    ppl::llm::utils::ParseTokens(a.Str("--stop-tokens"), &gc->stop_tokens);
    ppl::llm::utils::ParseTokens(a.Str("--special_tokens"), &gc->special_tokens);
    gc->max_cooldown_request = a.Int("--max-cooldown-request");
    gc->enable_prefix_cache = a.Bool("--enable-prefix-cache");
    gc->max_prefill_batch = gc->enable_prefix_cache ? 1 : a.Int("--max-prefill-batch");  // offline_inference.cc:97-99
    gc->enable_profiling = a.Bool("--enable-profiling");

    if (!ppl::llm::ParseModelConfig(rc->model_param_path, mc)) {
        std::cerr << "PaseModelConfig failed, model_param_path: " << rc->model_param_path << "\n";
        return false;
    }
    if (rc->tensor_parallel_size < 1 || (rc->tensor_parallel_size & (rc->tensor_parallel_size - 1))) {
        std::cerr << "tensor_parallel_size must be a power of two\n";  // offline_inference.cc:136-139
        return false;
    }
    return true;
}

typedef std::chrono::steady_clock Clock;

// in-process Connection: collects generated tokens and the time of the first / last response of every request
class LocalConnection final : public ppl::llm::Connection {
public:
    struct Rec {
        std::vector<int> tokens;
        std::string text;
        Clock::time_point submit, first, last;
        bool finished = false, failed = false;
    };
    void OnTokenize(uint64_t, const std::vector<int>&) override {}
    void OnProfiling(const std::shared_ptr<ppl::llm::WorkerProfiler>& p) override { ppl::llm::PrintProfiler(*p); }
    void MarkSubmit(uint64_t id) {
        std::lock_guard<std::mutex> g(mu_);
        recs_[id].submit = Clock::now();
    }
    void Send(const std::vector<ppl::llm::Response>& batch) override {
        const auto now = Clock::now();
        std::lock_guard<std::mutex> g(mu_);
        for (const auto& r : batch) {
            Rec& rec = recs_[r.id];
            if (rec.tokens.empty()) rec.first = now;
            rec.tokens.push_back(r.token);
            rec.text += r.generated;   // text requests: the detokenised piece(s) of this response
            rec.last = now;
            if (r.finish_flag != ppl::llm::FinishFlag::NOT_FINISHED) {
                rec.finished = true;
                ++count_;
            }
        }
        if (count_ >= wanted_) cv_.notify_all();
    }
    void NotifyFailure(uint64_t id, ppl::common::RetCode rc, const std::string& msg) override {
        std::lock_guard<std::mutex> g(mu_);
        std::cerr << "request " << id << " failed: " << ppl::common::GetRetCodeStr(rc) << " " << msg << "\n";
        recs_[id].failed = true;
        ++count_;
        if (count_ >= wanted_) cv_.notify_all();
    }
    void SetWanted(size_t n) {
        std::lock_guard<std::mutex> g(mu_);
        wanted_ = n;
        count_ = 0;
    }
    void Wait() {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return count_ >= wanted_; });
    }
    std::unordered_map<uint64_t, Rec>& records() { return recs_; }

private:
    std::mutex mu_;
    std::condition_variable cv_;
    size_t wanted_ = 0, count_ = 0;
    std::unordered_map<uint64_t, Rec> recs_;
};

inline double Ms(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }

inline double Percentile(std::vector<double> v, double p) {
    if (v.empty()) return 0;
    std::sort(v.begin(), v.end());
    size_t i = (size_t)(p / 100.0 * (v.size() - 1) + 0.5);
    return v[std::min(i, v.size() - 1)];
}

}  // namespace tools
