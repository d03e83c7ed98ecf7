// The ppl::nn::DeviceContext surface the reference uses (src/engine/llm_engine.h:140-142,
// src/backends/cuda/resource_manager.cc:183-211): a tensor's device, and the way to obtain the device's stream.
#pragma once
#include <stdint.h>

#include "ppl/common/retcode.h"

namespace ppl { namespace nn {

class DeviceContext {
public:
    virtual ~DeviceContext() {}
    virtual const char* GetType() const = 0;                               // "cpu" | "hip"
    virtual ppl::common::RetCode Configure(uint32_t option, ...) = 0;      // e.g. DEV_CONF_GET_STREAM
};

namespace llm { namespace cuda { enum { DEV_CONF_GET_STREAM = 0 }; }}

}}  // namespace ppl::nn
