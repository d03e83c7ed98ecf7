// benchmark_prefix_cache_offline: TTFT with and without the prefix cache (reference
// tools/benchmark_prefix_cache_offline.cc:298-513).  samples_8192.json is not in the reference tree
// (.MISSING_LARGE_BLOBS), so the long-context prompts are synthetic: --batch prompts of
// --prompt-len tokens whose first --shared-len tokens are common to all of them.  Like the reference (:442-508) the SAME prompt
// list is submitted twice to one generator:
//   run 1 ("first"):  the first prompt is cold; with --batch > 1 the others already hit the shared prefix it inserted
//                     (--max-prefill-batch 1 admits them one after another);
//   run 2 ("prefix"): every prompt is in the cache up to its last full page (cache-prefill kernel path).
// --second-run new-tails instead gives run 2 fresh unique tails behind the shared prefix (a partial hit of --shared-len tokens: the
// round-1/2 form of this tool).  Prints first ttft / prefix ttft / first_generate_time / prefix_generate_time like the reference:
// `ttft` is the reference's definition -- the FIRST Send of the run minus the begin of the run (:233-237, :454-460) -- and the
// JSON line adds the p50 / max over the requests of (first response - its own submit).
#include <fstream>
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
    a.Def("--dump-answers", "", "write the generated token ids (run 1 then run 2, one request per line) to this file");
    a.Def("--second-run", "same", "same: run 2 resubmits run 1's prompts (the reference); new-tails: fresh unique tails behind the shared prefix");
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

    const bool new_tails = a.Str("--second-run") == "new-tails";
    std::vector<std::shared_ptr<std::vector<int>>> prompts;
    auto make_prompts = [&]() {
        prompts.clear();
        for (int b = 0; b < batch; ++b) {
            auto t = std::make_shared<std::vector<int>>(shared);
            for (int i = slen; i < plen; ++i) t->push_back(tok(rng));
            prompts.push_back(t);
        }
    };
    make_prompts();
    double ttft[2] = {0, 0}, ttft_p50[2] = {0, 0}, ttft_max[2] = {0, 0}, total[2] = {0, 0};
    std::vector<std::vector<int>> answers[2];
    for (int run = 0; run < 2; ++run) {
        if (run == 1 && new_tails) make_prompts();
        std::vector<std::shared_ptr<Request>> reqs;
        for (int b = 0; b < batch; ++b) {
            auto r = std::make_shared<Request>((uint64_t)(run * batch + b), "", 1.0f, (uint32_t)gen_len);
            r->early_stopping = false;
            r->token_ids = std::make_shared<std::vector<int>>(*prompts[b]);
            reqs.push_back(r);
        }
        conn.SetWanted(reqs.size());
        const auto t0 = tools::Clock::now();
        for (auto& r : reqs) { conn.MarkSubmit(r->id); generator->Process(r); }
        conn.Wait();
        const auto t1 = tools::Clock::now();
        std::vector<double> v;
        auto first_send = tools::Clock::time_point::max();
        for (auto& r : reqs) {
            const auto& rec = conn.records()[r->id];
            if (rec.failed) {  // e.g. prompt longer than --max-input-tokens-per-request
                std::cerr << "run " << run << ": request " << r->id << " was rejected; no timing reported\n";
                return 1;
            }
            v.push_back(tools::Ms(rec.submit, rec.first));
            first_send = std::min(first_send, rec.first);
            answers[run].push_back(rec.tokens);
        }
        ttft[run] = tools::Ms(t0, first_send);  // the reference's number: first Send of the run - begin
        ttft_p50[run] = tools::Percentile(v, 50);
        ttft_max[run] = tools::Percentile(v, 100);
        total[run] = tools::Ms(t0, t1);
        while (!generator->IsIdle()) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    // same prompts, greedy: the cached run must answer what the cold run answered (compared outside rounding-noise ties by
    // tests/test_gpu_tools.py, which reads the token lists from --dump-answers)
    int same_answers = 0;
    if (!new_tails)
        for (int b = 0; b < batch; ++b) same_answers += answers[0][b] == answers[1][b];
    std::cout << "first ttft: " << ttft[0] << " ms\nprefix ttft: " << ttft[1] << " ms\nfirst_generate_time: " << total[0]
              << " ms\nprefix_generate_time: " << total[1] << " ms" << std::endl;
    char buf[1024];
    snprintf(buf, sizeof(buf), "{\"prompt_len\":%d,\"shared_len\":%d,\"batch\":%d,\"generation_length\":%d,\"second_run\":\"%s\","
             "\"first_ttft_ms\":%.3f,\"prefix_ttft_ms\":%.3f,\"first_ttft_p50_ms\":%.3f,\"prefix_ttft_p50_ms\":%.3f,"
             "\"first_ttft_max_ms\":%.3f,\"prefix_ttft_max_ms\":%.3f,\"first_generate_ms\":%.3f,\"prefix_generate_ms\":%.3f,"
             "\"identical_answers\":%d}", plen, slen, batch, gen_len, new_tails ? "new-tails" : "same", ttft[0], ttft[1], ttft_p50[0],
             ttft_p50[1], ttft_max[0], ttft_max[1], total[0], total[1], new_tails ? -1 : same_answers);
    std::cout << buf << std::endl;
    if (!a.Str("--dump-answers").empty()) {
        std::ofstream f(a.Str("--dump-answers"));
        for (int run = 0; run < 2; ++run)
            for (auto& t : answers[run]) {
                for (size_t i = 0; i < t.size(); ++i) f << (i ? " " : "") << t[i];
                f << "\n";
            }
    }
    generator.reset();
    return 0;
}
