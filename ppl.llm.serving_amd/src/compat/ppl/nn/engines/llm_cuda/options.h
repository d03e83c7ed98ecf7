// Engine options the reference names (src/engine/llm_engine.cc:114, src/backends/cuda/resource_manager.cc:43-112).
#pragma once
#include <stdint.h>

namespace ppl { namespace nn { namespace llm { namespace cuda {

enum {
    ENGINE_CONF_CACHE_PREFILL = 0,        // int: 1 = this step starts behind cached KV (prefix-cache hit)
    ENGINE_CONF_DECODING_SHM_MHA, ENGINE_CONF_DECODING_INF_MHA, ENGINE_CONF_DECODING_INF_GQA,
    ENGINE_CONF_DECODING_ATTN_SPLIT_K, ENGINE_CONF_DECODING_ATTN_TPB, ENGINE_CONF_GRAPH_FUSION,
    ENGINE_CONF_MAX,
};

}}}}  // namespace ppl::nn::llm::cuda
