// Boundary proof for SURVEY.md 8(b) B1/B2 ON THE DEVICE (TEST INFRASTRUCTURE; built by `make ref` in the build container, the
// binary travels to the GPU box with the snapshot).
//
// The REFERENCE'S OWN llm_generator.cc + llm_engine.cc (compiled in place from /root/reference/src, unmodified) drive
// libpplhip.so through the ppl::nn::Runtime / Tensor / Engine objects of ppl.llm.serving_amd/src/backends/hip_nn:
// the engine binds 11 tensors by index and copies them per step exactly as it does over ppl.nn's CUDA engine.
// Workload = the reference's offline smoke run (tools/offline_inference.cc:304-309,376-413) as token ids: 4 fixed prompts,
// generation_length 8 + i.  Output format = ppl.llm.serving_amd/tools/offline_inference --workload prompts4, whose tokens
// (the repo's own generator + engine over the same libpplhip) tests/test_gpu_tools.py requires to be IDENTICAL.
//
//   ref_backend_driver <params.json> [tensor_parallel_size [scenario.json]]
// With a scenario file (tools/scenario.h) the requests, sampling parameters and generator limits come from it and the answers are
// printed as offline_inference --workload scenario prints them.
#include <condition_variable>
#include <fstream>
#include <iostream>
#include <mutex>
#include <sstream>

#include "generator/llm_generator.h"                                   // the reference's
#include "../../ppl.llm.serving_amd/src/backends/hip_nn/hip_nn_backend.h"
#include "../../ppl.llm.serving_amd/src/utils/mini_json.h"
#include "../../ppl.llm.serving_amd/tools/scenario.h"

using namespace ppl::llm;
using namespace ppl::common;

namespace {
class Conn final : public Connection {
public:
    void OnProfiling(const std::shared_ptr<WorkerProfiler>&) override {}
    void OnTokenize(uint64_t, const std::vector<int>&) override {}
    void Send(const std::vector<Response>& rsps) override {
        std::lock_guard<std::mutex> g(mu_);
        for (const auto& r : rsps) {
            tokens[r.id].push_back(r.token);
            if (r.finish_flag != FinishFlag::NOT_FINISHED) ++done;
        }
        cv_.notify_all();
    }
    void NotifyFailure(uint64_t id, RetCode, const std::string& msg) override {
        std::lock_guard<std::mutex> g(mu_);
        std::cerr << "request " << id << " failed: " << msg << "\n";
        failed.push_back(id);
        ++done;
        cv_.notify_all();
    }
    bool Wait(size_t n, int ms) {
        std::unique_lock<std::mutex> lk(mu_);
        return cv_.wait_for(lk, std::chrono::milliseconds(ms), [&] { return done >= n; });
    }
    std::map<uint64_t, std::vector<int>> tokens;
    std::vector<uint64_t> failed;
    size_t done = 0;

private:
    std::mutex mu_;
    std::condition_variable cv_;
};
}  // namespace

int main(int argc, char** argv) {
    if (argc < 2) {
        std::cerr << "usage: ref_backend_driver <params.json> [tensor_parallel_size]\n";
        return 2;
    }
    std::ifstream ifs(argv[1]);
    std::stringstream buf;
    buf << ifs.rdbuf();
    utils::JsonValue doc;
    if (!utils::JsonParser(buf.str()).Parse(&doc)) {
        std::cerr << "bad params.json\n";
        return 2;
    }
    ModelConfig mc;  // the reference's struct (src/common/config.h); its ParseModelConfig needs rapidjson, absent here
    mc.hidden_dim = (int32_t)doc.GetInt("hidden_dim", 0);
    mc.intermediate_dim = (int32_t)doc.GetInt("intermediate_dim", 0);
    mc.num_layers = (int32_t)doc.GetInt("num_layers", 0);
    mc.num_heads = (int32_t)doc.GetInt("num_heads", 0);
    mc.num_kv_heads = (int32_t)doc.GetInt("num_kv_heads", mc.num_heads);
    mc.vocab_size = (int32_t)doc.GetInt("vocab_size", 0);
    mc.cache_quant_bit = (int32_t)doc.GetInt("cache_quant_bit", 0);
    mc.cache_quant_group = (int32_t)doc.GetInt("cache_quant_group", 1);
    mc.cache_layout = (int32_t)doc.GetInt("cache_layout", 0);
    mc.cache_mode = (int32_t)doc.GetInt("cache_mode", 0);
    mc.page_size = (int32_t)doc.GetInt("page_size", 0);
    mc.dynamic_batching = doc.GetBool("dynamic_batching", true);
    mc.auto_causal = doc.GetBool("auto_causal", true);
    hip_nn::ExtraConfig ex;
    ex.weight_quant_bit = (int32_t)doc.GetInt("weight_quant_bit", 0);
    ex.weight_quant_group = (int32_t)doc.GetInt("weight_quant_group", 128);
    ex.max_position = (int32_t)doc.GetInt("max_position", 4096);
    ex.synthetic_weights = true;
    ex.synthetic_seed = 1234;                       // tools' --synthetic-seed default
    ex.kv_cache_max_tokens = 8192;
    ResourceConfig rc;
    rc.tensor_parallel_size = argc > 2 ? atoi(argv[2]) : 1;
    rc.max_tokens_scale = 0.94f;
    rc.max_running_batch = 1024;
    GeneratorConfig gc;                             // defaults of tools/offline_inference.cc:40-90
    gc.top_k = 1;
    gc.max_running_batch = 1024;
    gc.max_input_tokens_per_request = 4096;
    gc.max_output_tokens_per_request = 4096;
    gc.max_total_tokens_per_request = 8192;
    gc.max_tokens_per_step = 8192;
    gc.max_prefill_batch = 64;
    utils::JsonValue scn;
    const bool have_scenario = argc > 3;
    if (have_scenario) {
        if (!scenario::LoadScenario(argv[3], &scn)) {
            std::cerr << "cannot read scenario " << argv[3] << "\n";
            return 2;
        }
        scenario::ScenarioGeneratorConfig(scn, &gc);
        rc.enable_penalty = gc.enable_penalty;
        rc.max_running_batch = gc.max_running_batch;
        ex.max_tokens_per_step = gc.max_tokens_per_step;
        if (scn.Find("kv_cache_max_tokens")) ex.kv_cache_max_tokens = (uint64_t)scn.GetInt("kv_cache_max_tokens", 8192);
    }

    hip_nn::Backend backend;
    if (backend.Init(mc, rc, ex) != RC_SUCCESS) {
        std::cerr << "backend init failed\n";
        return 1;
    }
    Resource resource;
    backend.FillResource(&resource);
    Conn conn;
    const std::vector<std::vector<int>> prompts = {{1, 15043, 29892, 590, 1024, 338}, {1, 450, 6673, 310, 278, 3303, 3900, 338},
                                                   {1, 450, 7483, 310, 3444, 338}, {1, 450, 5434, 310, 319, 29902, 338}};
    {
        LLMGenerator gen(resource, gc, mc, &conn);
        if (gen.Init() != RC_SUCCESS) {
            std::cerr << "generator init failed\n";
            return 1;
        }
        std::vector<std::shared_ptr<Request>> reqs;
        if (have_scenario) {
            reqs = scenario::ScenarioRequests<Request>(scn, mc.vocab_size);
            srand(1);  // as offline_inference --workload scenario: both processes then draw the same rand() numbers in the sampler
            for (auto& r : reqs) gen.Process(r);
            if (!conn.Wait(reqs.size(), 300000)) {
                std::cerr << "timed out\n";
                return 1;
            }
            for (uint64_t id : conn.failed) conn.tokens.erase(id);
            scenario::PrintScenarioResult(conn.tokens, conn.failed);
            return 0;
        }
        for (size_t i = 0; i < prompts.size(); ++i) {
            auto r = std::make_shared<Request>(i, "", 1.0f, 8 + (uint32_t)i);
            r->token_ids = std::make_shared<std::vector<int>>();
            for (int t : prompts[i]) r->token_ids->push_back(t % mc.vocab_size);
            reqs.push_back(r);
            gen.Process(r);
        }
        if (!conn.Wait(prompts.size(), 120000)) {
            std::cerr << "timed out\n";
            return 1;
        }
        for (auto& r : reqs) {
            std::cout << "Prompt tokens:";
            for (int t : *r->token_ids) std::cout << " " << t;
            std::cout << "\nAnswer tokens:";
            for (int t : conn.tokens[r->id]) std::cout << " " << t;
            std::cout << "\n";
        }
    }   // generator before backend (ownership rule, offline_inference.cc:414)
    return 0;
}
