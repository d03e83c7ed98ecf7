// A request scenario for differential runs (tests/test_gpu_tools.py): the same JSON drives this tree's offline_inference
// (--workload scenario --scenario-file f) and tests/host/ref_backend_driver (the reference's generator + engine).
//   {"generator": {"max_running_batch", "max_tokens_per_step", "max_prefill_batch", "max_cooldown_request", "enable_prefix_cache",
//                  "enable_penalty", "stop_tokens": [..]},
//    "kv_cache_max_tokens": N,
//    "requests": [{"id", "tokens": [..], "generation_length", "temperature", "top_p", "top_k", "repetition_penalty",
//                  "presence_penalty", "frequency_penalty", "early_stopping", "stop_tokens": [..]}]}
// Templates: Request / GeneratorConfig are this tree's or the reference's types (same member names by construction).
#pragma once
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <unordered_set>
#include <vector>

#include "../src/utils/mini_json.h"

namespace ppl { namespace llm { namespace scenario {

inline bool LoadScenario(const std::string& path, utils::JsonValue* doc) {
    std::ifstream ifs(path);
    if (!ifs.is_open()) return false;
    std::stringstream buf;
    buf << ifs.rdbuf();
    return utils::JsonParser(buf.str()).Parse(doc) && doc->Find("requests") != nullptr;
}

template <typename GeneratorConfigT>
void ScenarioGeneratorConfig(const utils::JsonValue& doc, GeneratorConfigT* gc) {
    const utils::JsonValue* g = doc.Find("generator");
    if (!g) return;
    gc->max_running_batch = (int)g->GetInt("max_running_batch", gc->max_running_batch);
    gc->max_tokens_per_step = (int)g->GetInt("max_tokens_per_step", gc->max_tokens_per_step);
    gc->max_cooldown_request = (int)g->GetInt("max_cooldown_request", gc->max_cooldown_request);
    gc->enable_prefix_cache = g->GetBool("enable_prefix_cache", gc->enable_prefix_cache);
    gc->enable_penalty = g->GetBool("enable_penalty", gc->enable_penalty);
    gc->max_prefill_batch = gc->enable_prefix_cache ? 1 : (int)g->GetInt("max_prefill_batch", gc->max_prefill_batch);  // offline_inference.cc:97-99
    if (const utils::JsonValue* st = g->Find("stop_tokens"))
        for (const auto& t : st->arr) gc->stop_tokens.insert((int)t.AsInt());
}

template <typename RequestT>
std::vector<std::shared_ptr<RequestT>> ScenarioRequests(const utils::JsonValue& doc, int vocab_size) {
    std::vector<std::shared_ptr<RequestT>> out;
    for (const auto& r : doc.Find("requests")->arr) {
        auto q = std::make_shared<RequestT>((uint64_t)r.GetInt("id", (int64_t)out.size()), "", (float)r.GetNum("temperature", 1.0),
                                            (uint32_t)r.GetInt("generation_length", 8));
        q->top_p = (float)r.GetNum("top_p", 0.0);
        q->top_k = (int)r.GetInt("top_k", 1);
        q->repetition_penalty = (float)r.GetNum("repetition_penalty", 1.0);
        q->presence_penalty = (float)r.GetNum("presence_penalty", 0.0);
        q->frequency_penalty = (float)r.GetNum("frequency_penalty", 0.0);
        q->early_stopping = r.GetBool("early_stopping", true);
        q->token_ids = std::make_shared<std::vector<int>>();
        if (const utils::JsonValue* t = r.Find("tokens"))
            for (const auto& v : t->arr) q->token_ids->push_back((int)(v.AsInt() % vocab_size));
        q->stop_tokens = std::make_shared<std::unordered_set<int>>();
        if (const utils::JsonValue* st = r.Find("stop_tokens"))
            for (const auto& v : st->arr) q->stop_tokens->insert((int)v.AsInt());
        out.push_back(q);
    }
    return out;
}

// {"tokens": {"id": [..]}, "failed": [ids]}
template <typename TokMap, typename FailedVec>
void PrintScenarioResult(const TokMap& tokens, const FailedVec& failed) {
    std::cout << "{\"tokens\":{";
    bool first = true;
    for (const auto& kv : tokens) {
        std::cout << (first ? "" : ",") << "\"" << kv.first << "\":[";
        for (size_t i = 0; i < kv.second.size(); ++i) std::cout << (i ? "," : "") << kv.second[i];
        std::cout << "]";
        first = false;
    }
    std::cout << "},\"failed\":[";
    for (size_t i = 0; i < failed.size(); ++i) std::cout << (i ? "," : "") << failed[i];
    std::cout << "]}" << std::endl;
}

}}}  // namespace ppl::llm::scenario
