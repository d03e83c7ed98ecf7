// benchmark_prefix_cache_offline: TTFT with and without the prefix cache (reference
// tools/benchmark_prefix_cache_offline.cc:298-513).  samples_8192.json is not in the reference tree
// (.MISSING_LARGE_BLOBS), so the long-context request is synthetic: --prompt-len tokens of which the first
// --shared-len are common to both runs.  The same generator instance serves
//   run 1 ("first"):  prompt = shared prefix + unique tail A   -> cold, its full pages enter the prefix cache
//   run 2 ("prefix"): prompt = shared prefix + unique tail B   -> hits the cached pages (cache-prefill kernel path)
// and prints first ttft / prefix ttft / first_generate_time / prefix_generate_time like the reference (:504-508).
#include <iostream>
#include <random>
#include <thread>

#include "tool_common.h"

using namespace ppl::llm;
using namespace ppl::common;

int main(int argc, char** argv) {
    tools::Args a;
    tools::DefineCommonFlags(&a);
    a.Def("--prompt-len", "8192", "tokens per prompt");
    a.Def("--shared-len", "6144", "length of the prefix shared by the two runs");
    a.Def("--generation-length", "32", "benchmark_prefix_cache_offline.cc:443");
    a.Def("--batch", "1", "requests per run");
    if (!a.Parse(argc, argv)) return -1;
    if (a.Bool("--help")) { a.PrintHelp(); return 0; }

    ResourceConfig rc;
    GeneratorConfig gc;
    ModelConfig mc;
    if (!tools::FillConfigs(a, &rc, &gc, &mc)) return -1;
    if (!gc.enable_prefix_cache) std::cerr << "note: --enable-prefix-cache is off, both runs will be cold\n";
    if (mc.cache_mode != 1) { std::cerr << "the prefix cache needs cache_mode 1 (paged KV)\n"; return -1; }

    hip::HipResourceManager resource_manager;
    RetCode st = resource_manager.Init(mc, rc);
    if (st != RC_SUCCESS) { std::cerr << "init HipResourceManager failed: " << GetRetCodeStr(st) << "\n"; return -1; }
    Resource resource;
    resource_manager.FillResource(&resource);

    const int plen = a.Int("--prompt-len"), slen = std::min(a.Int("--shared-len"), a.Int("--prompt-len"));
    const int batch = a.Int("--batch"), gen_len = a.Int("--generation-length");
    std::mt19937_64 rng((uint64_t)a.I64("--seed"));
    std::uniform_int_distribution<int> tok(3, mc.vocab_size - 1);
    std::vector<int> shared(slen);
    for (int& t : shared) t = tok(rng);

    tools::LocalConnection conn;
    auto generator = std::make_unique<LLMGenerator>(resource, gc, mc, &conn);
    st = generator->Init();
    if (st != RC_SUCCESS) { std::cerr << "llm_generator init failed: " << GetRetCodeStr(st) << "\n"; return -1; }

    double ttft[2] = {0, 0}, total[2] = {0, 0};
    for (int run = 0; run < 2; ++run) {
        std::vector<std::shared_ptr<Request>> reqs;
        for (int b = 0; b < batch; ++b) {
            auto r = std::make_shared<Request>((uint64_t)(run * batch + b), "", 1.0f, (uint32_t)gen_len);
            r->early_stopping = false;
            r->token_ids = std::make_shared<std::vector<int>>(shared);
            for (int i = slen; i < plen; ++i) r->token_ids->push_back(tok(rng));
            reqs.push_back(r);
        }
        conn.SetWanted(reqs.size());
        const auto t0 = tools::Clock::now();
        for (auto& r : reqs) { conn.MarkSubmit(r->id); generator->Process(r); }
        conn.Wait();
        const auto t1 = tools::Clock::now();
        std::vector<double> v;
        for (auto& r : reqs) {
            if (conn.records()[r->id].failed) {  // e.g. prompt longer than --max-input-tokens-per-request
                std::cerr << "run " << run << ": request " << r->id << " was rejected; no timing reported\n";
                return 1;
            }
            v.push_back(tools::Ms(conn.records()[r->id].submit, conn.records()[r->id].first));
        }
        ttft[run] = tools::Percentile(v, 50);
        total[run] = tools::Ms(t0, t1);
        while (!generator->IsIdle()) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    std::cout << "first ttft: " << ttft[0] << " ms\nprefix ttft: " << ttft[1] << " ms\nfirst_generate_time: " << total[0]
              << " ms\nprefix_generate_time: " << total[1] << " ms" << std::endl;
    char buf[512];
    snprintf(buf, sizeof(buf), "{\"prompt_len\":%d,\"shared_len\":%d,\"batch\":%d,\"generation_length\":%d,\"first_ttft_ms\":%.3f,"
             "\"prefix_ttft_ms\":%.3f,\"first_generate_ms\":%.3f,\"prefix_generate_ms\":%.3f}", plen, slen, batch, gen_len, ttft[0],
             ttft[1], total[0], total[1]);
    std::cout << buf << std::endl;
    generator.reset();
    return 0;
}
