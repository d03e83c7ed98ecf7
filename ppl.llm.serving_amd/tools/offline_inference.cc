// offline_inference: the in-process driver of the hot path (reference tools/offline_inference.cc:303-417) on the
// hip backend.
//   --workload prompts4     the reference's smoke run: 4 fixed prompts, generation_length 8 + i, prints the answers and the
//                           generation time (offline_inference.cc:304-309,376-413).  With --tokenizer-path the prompts are the
//                           reference's TEXTS, tokenised / detokenised by src/tokenizer; without it their token-id stand-ins
//   --workload samples1024  a samples_1024.json-shaped token-in/out load (the reference's client_qps_measure workload,
//                           tools/client_qps_measure.cc:54-96): N requests, log-normal prompt / answer lengths, EOS
//                           ignored; prints ONE JSON line with tokens/s and the TTFT distribution
//                           (metric definitions: client_qps_measure.cc:285-287,331,337-340)
#include <algorithm>
#include <iostream>
#include <random>
#include <thread>

#include "tool_common.h"
#include <map>
#include "tokenizer/tokenizer_factory.h"
#include "scenario.h"

using namespace ppl::llm;
using namespace ppl::common;

int main(int argc, char** argv) {
    tools::Args a;
    tools::DefineCommonFlags(&a);
    a.Def("--workload", "prompts4", "prompts4 | samples1024 | scenario");
    a.Def("--scenario-file", "", "scenario: request / generator description (tools/scenario.h), answers printed as one JSON line");
    a.Def("--num-requests", "1024", "samples1024: number of requests");
    a.Def("--max-seq-len", "1024", "samples1024: prompt + answer length cap (seqlen of the benchmark config)");
    a.Def("--request-rate", "0", "samples1024: Poisson arrivals at this many requests per second (0: all submitted at once), as "
                                 "client_qps_measure's --request_rate");
    if (!a.Parse(argc, argv)) return -1;
    if (a.Bool("--help")) { a.PrintHelp(); return 0; }

    ResourceConfig rc;
    GeneratorConfig gc;
    ModelConfig mc;
    if (!tools::FillConfigs(a, &rc, &gc, &mc)) return -1;
    utils::JsonValue scn;
    if (a.Str("--workload") == "scenario") {
        if (!scenario::LoadScenario(a.Str("--scenario-file"), &scn)) {
            std::cerr << "cannot read --scenario-file " << a.Str("--scenario-file") << "\n";
            return -1;
        }
        scenario::ScenarioGeneratorConfig(scn, &gc);
        rc.enable_penalty = gc.enable_penalty;
        rc.max_running_batch = gc.max_running_batch;
        rc.max_tokens_per_step = gc.max_tokens_per_step;
        if (scn.Find("kv_cache_max_tokens")) rc.kv_cache_max_tokens_override = (uint64_t)scn.GetInt("kv_cache_max_tokens", 0);
    }

    hip::HipResourceManager resource_manager;
    RetCode st = resource_manager.Init(mc, rc);
    if (st != RC_SUCCESS) {
        std::cerr << "init HipResourceManager failed: " << GetRetCodeStr(st) << "\n";
        return -1;
    }
    Resource resource;
    resource_manager.FillResource(&resource);
    std::unique_ptr<Tokenizer> tokenizer;
    if (!a.Str("--tokenizer-path").empty()) {   // offline_inference.cc:351-358
        tokenizer.reset(TokenizerFactory::Create(a.Str("--model-type"), a.Str("--tokenizer-type"), a.Str("--tokenizer-path"),
                                                 a.Str("--tokenizer-config-path")));
        if (!tokenizer) {
            std::cerr << "create tokenizer failed\n";
            return -1;
        }
        resource.tokenizer = tokenizer.get();
    }

    // ---- the requests ------------------------------------------------------------------------------------
    std::vector<std::shared_ptr<Request>> requests;
    std::mt19937_64 rng((uint64_t)a.I64("--seed"));
    const std::string workload = a.Str("--workload");
    if (workload == "prompts4") {
        // token-id stand-ins for "Hello, my name is" / "The president of the United States is" / ... (BOS = 1 first)
        const std::vector<std::vector<int>> prompts = {
            {1, 15043, 29892, 590, 1024, 338}, {1, 450, 6673, 310, 278, 3303, 3900, 338},
            {1, 450, 7483, 310, 3444, 338}, {1, 450, 5434, 310, 319, 29902, 338}};
        static const char* texts[4] = {"Hello, my name is", "The president of the United States is", "The capital of France is",
                                       "The future of AI is"};   // offline_inference.cc:304-309
        for (size_t i = 0; i < prompts.size(); ++i) {
            auto r = std::make_shared<Request>(i, tokenizer ? texts[i] : "", 1.0f, 8 + (uint32_t)i);
            if (!tokenizer) {
                r->token_ids = std::make_shared<std::vector<int>>();
                for (int t : prompts[i]) r->token_ids->push_back(t % mc.vocab_size);
            }
            requests.push_back(r);
        }
    } else if (workload == "samples1024") {
        // lengths shaped like tools/samples_1024.json at ~4 chars/token: prompt median 30 / mean 104,
        // answer median 286 / mean 315, both clipped to [4, 1024]; prompt + answer <= --max-seq-len
        std::lognormal_distribution<double> plen(std::log(30.0), 1.577), olen(std::log(286.0), 0.44);
        std::uniform_int_distribution<int> tok(3, mc.vocab_size - 1);
        const int n = a.Int("--num-requests"), cap = a.Int("--max-seq-len");
        for (int i = 0; i < n; ++i) {
            int p = std::min(1024, std::max(4, (int)plen(rng)));
            int o = std::min(1024, std::max(4, (int)olen(rng)));
            if (p > cap - 4) p = cap - 4;
            if (p + o > cap) o = cap - p;
            auto r = std::make_shared<Request>((uint64_t)i, "", 1.0f, (uint32_t)o);
            r->early_stopping = false;  // EOS ignored (client_qps_measure.cc:88)
            r->token_ids = std::make_shared<std::vector<int>>(p);
            for (int& t : *r->token_ids) t = tok(rng);
            requests.push_back(r);
        }
    } else if (workload == "scenario") {
        requests = scenario::ScenarioRequests<Request>(scn, mc.vocab_size);
    } else {
        std::cerr << "unknown --workload " << workload << "\n";
        return -1;
    }

    tools::LocalConnection conn;
    conn.SetWanted(requests.size());
    auto generator = std::make_unique<LLMGenerator>(resource, gc, mc, &conn);
    st = generator->Init();
    if (st != RC_SUCCESS) {
        std::cerr << "llm_generator init failed: " << GetRetCodeStr(st) << "\n";
        return -1;
    }

    // scenario runs are compared token for token with another process (tests/test_gpu_tools.py): the sampler draws from the C
    // library's rand() like the reference (post_processor.cc:179-183, never seeded), and start-up code of the runtime libraries
    // consumes an unpredictable number of values first -- restart the sequence here so that both processes draw the same numbers
    if (workload == "scenario") srand(1);
    uint64_t generate_us = 0;
    const auto t_begin = tools::Clock::now();
    {
        utils::TimingGuard timing(&generate_us);
        const double rate = a.Num("--request-rate");
        std::exponential_distribution<double> gap(rate > 0 ? rate : 1.0);
        auto next = tools::Clock::now();
        for (auto& r : requests) {
            if (rate > 0 && workload == "samples1024") {  // client_qps_measure.cc: exponential inter-arrival times
                std::this_thread::sleep_until(next);
                next += std::chrono::duration_cast<tools::Clock::duration>(std::chrono::duration<double>(gap(rng)));
            }
            conn.MarkSubmit(r->id);
            generator->Process(r);
        }
        conn.Wait();
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(50));

    if (workload == "prompts4") {
        for (auto& r : requests) {
            if (tokenizer) std::cout << "Prompt: " << r->prompt << "\nAnswer: " << conn.records()[r->id].text << "\n";
            std::cout << "Prompt tokens:";
            for (int t : *r->token_ids) std::cout << " " << t;
            std::cout << "\nAnswer tokens:";
            for (int t : conn.records()[r->id].tokens) std::cout << " " << t;
            std::cout << "\n";
        }
        std::cout << "generation time: " << generate_us / 1e3 << "ms" << std::endl;
    } else if (workload == "scenario") {
        std::map<uint64_t, std::vector<int>> tokens;
        std::vector<uint64_t> failed;
        for (auto& r : requests) {
            auto& rec = conn.records()[r->id];
            if (rec.failed) failed.push_back(r->id);
            else tokens[r->id] = rec.tokens;
        }
        scenario::PrintScenarioResult(tokens, failed);
    } else {
        std::vector<double> ttft, tpot;
        uint64_t out_tokens = 0, in_tokens = 0, failed = 0;
        tools::Clock::time_point last = t_begin, first_any = tools::Clock::time_point::max();
        for (auto& r : requests) {
            auto& rec = conn.records()[r->id];
            if (rec.failed || rec.tokens.empty()) { ++failed; continue; }
            ttft.push_back(tools::Ms(rec.submit, rec.first));
            if (rec.tokens.size() > 1) tpot.push_back(tools::Ms(rec.first, rec.last) / (rec.tokens.size() - 1));
            out_tokens += rec.tokens.size();
            in_tokens += r->token_ids->size();
            last = std::max(last, rec.last);
            first_any = std::min(first_any, rec.first);
        }
        const double wall_s = tools::Ms(t_begin, last) / 1e3;
        const auto& prof = generator->GetProfiler();
        const auto& g = prof.step_counter.global;
        char buf[2048];
        snprintf(buf, sizeof(buf),
                 "{\"workload\":\"samples1024-shaped token-in/out, %zu requests, seed %lld\",\"request_rate\":%g,\"requests\":%zu,\"failed\":%llu,"
                 "\"input_tokens\":%llu,\"output_tokens\":%llu,\"wall_s\":%.4f,\"tokens_out_per_s\":%.2f,"
                 "\"generator_tps\":%.2f,\"steps\":%llu,\"max_running\":%llu,"
                 "\"ttft_ms\":{\"min\":%.2f,\"p10\":%.2f,\"p25\":%.2f,\"p50\":%.2f,\"p75\":%.2f,\"p90\":%.2f,\"p99\":%.2f,\"max\":%.2f},"
                 "\"decode_ms_per_token\":{\"p50\":%.3f,\"p90\":%.3f,\"p99\":%.3f},"
                 "\"phase_ms\":{\"prepare\":%.1f,\"set_input\":%.1f,\"model_forward\":%.1f,\"choose_token\":%.1f,\"post_process\":%.1f,\"total\":%.1f}}",
                 requests.size(), a.I64("--seed"), a.Num("--request-rate"), requests.size(), (unsigned long long)failed, (unsigned long long)in_tokens,
                 (unsigned long long)out_tokens, wall_s, out_tokens / wall_s,
                 g.total_cost ? g.output_token_cnt / (g.total_cost / 1e6) : 0.0, (unsigned long long)g.step_cnt,
                 (unsigned long long)prof.max_running_task, tools::Percentile(ttft, 0), tools::Percentile(ttft, 10),
                 tools::Percentile(ttft, 25), tools::Percentile(ttft, 50), tools::Percentile(ttft, 75), tools::Percentile(ttft, 90),
                 tools::Percentile(ttft, 99), tools::Percentile(ttft, 100), tools::Percentile(tpot, 50), tools::Percentile(tpot, 90),
                 tools::Percentile(tpot, 99), g.prepare_cost / 1e3, g.set_input_cost / 1e3, g.model_forward_cost / 1e3,
                 g.choose_token_cost / 1e3, g.post_process_cost / 1e3, g.total_cost / 1e3);
        std::cout << buf << std::endl;
    }
    generator.reset();  // before the resource manager (ownership rule, offline_inference.cc:414)
    return 0;
}
