// The ppl::nn::Engine surface the reference uses (src/engine/llm_engine.cc:114: Configure(ENGINE_CONF_CACHE_PREFILL, 0|1)).
#pragma once
#include <stdint.h>

#include "ppl/common/retcode.h"

namespace ppl { namespace nn {

class Engine {
public:
    virtual ~Engine() {}
    virtual const char* GetName() const = 0;
    virtual ppl::common::RetCode Configure(uint32_t option, ...) = 0;
};

}}  // namespace ppl::nn
