// params.json -> ModelConfig.  Same keys, same required/optional split and the same failure behaviour as the
// reference parser (src/common/config.cc:31-148): a missing mandatory key logs an error and fails; "num_kv_heads"
// defaults to num_heads; "page_size" is mandatory only when cache_mode == 1.
#include "config.h"

#include <fstream>
#include <sstream>

#include "../utils/mini_json.h"
#include "ppl/common/log.h"

namespace ppl { namespace llm {

namespace {

bool NeedInt(const utils::JsonValue& doc, const char* key, int32_t* out) {
    const utils::JsonValue* v = doc.Find(key);
    if (!v || v->type != utils::JsonValue::NUMBER) {
        LOG(ERROR) << "find key [" << key << "] failed";
        return false;
    }
    *out = (int32_t)v->AsInt();
    LOG(INFO) << "model_config." << key << ": " << *out;
    return true;
}

bool NeedBool(const utils::JsonValue& doc, const char* key, bool* out) {
    const utils::JsonValue* v = doc.Find(key);
    if (!v || (v->type != utils::JsonValue::BOOL && v->type != utils::JsonValue::NUMBER)) {
        LOG(ERROR) << "find key [" << key << "] failed";
        return false;
    }
    *out = v->type == utils::JsonValue::BOOL ? v->b : (v->num != 0);
    LOG(INFO) << "model_config." << key << ": " << *out;
    return true;
}

}  // namespace

bool ParseModelConfigFromString(const std::string& text, ModelConfig* mc) {
    utils::JsonValue doc;
    if (!utils::JsonParser(text).Parse(&doc) || doc.type != utils::JsonValue::OBJECT) {
        LOG(ERROR) << "ParseStream failed";
        return false;
    }
    if (!NeedInt(doc, "num_heads", &mc->num_heads)) return false;
    if (const utils::JsonValue* v = doc.Find("num_kv_heads")) mc->num_kv_heads = (int32_t)v->AsInt();
    else mc->num_kv_heads = mc->num_heads;
    if (!NeedInt(doc, "num_layers", &mc->num_layers)) return false;
    if (!NeedInt(doc, "hidden_dim", &mc->hidden_dim)) return false;
    if (!NeedInt(doc, "intermediate_dim", &mc->intermediate_dim)) return false;
    if (!NeedInt(doc, "vocab_size", &mc->vocab_size)) return false;
    if (!NeedInt(doc, "cache_quant_bit", &mc->cache_quant_bit)) return false;
    if (!NeedInt(doc, "cache_quant_group", &mc->cache_quant_group)) return false;
    if (!NeedInt(doc, "cache_layout", &mc->cache_layout)) return false;
    if (!NeedInt(doc, "cache_mode", &mc->cache_mode)) return false;
    if (mc->cache_mode == 1 && !NeedInt(doc, "page_size", &mc->page_size)) return false;
    if (!NeedBool(doc, "dynamic_batching", &mc->dynamic_batching)) return false;
    if (!NeedBool(doc, "auto_causal", &mc->auto_causal)) return false;
    // optional keys of this build (the reference keeps these inside the exported graph)
    mc->norm_eps = (float)doc.GetNum("norm_eps", mc->norm_eps);
    mc->rope_theta = (float)doc.GetNum("rope_theta", mc->rope_theta);
    mc->max_position = (int32_t)doc.GetInt("max_position", mc->max_position);
    mc->weight_quant_bit = (int32_t)doc.GetInt("weight_quant_bit", mc->weight_quant_bit);
    mc->weight_quant_group = (int32_t)doc.GetInt("weight_quant_group", mc->weight_quant_group);
    return true;
}

bool ParseModelConfig(const std::string& path, ModelConfig* mc) {
    std::ifstream ifs(path);
    if (!ifs.is_open()) {
        LOG(ERROR) << "cannot open [" << path << "]";
        return false;
    }
    std::stringstream ss;
    ss << ifs.rdbuf();
    return ParseModelConfigFromString(ss.str(), mc);
}

}}  // namespace ppl::llm
