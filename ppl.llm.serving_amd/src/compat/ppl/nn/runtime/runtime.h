// The ppl::nn::Runtime surface the reference drives (src/engine/llm_engine.h:124-138 binds inputs
// 0..10 and output 0 by index, src/engine/llm_engine.cc:115 runs it).  Abstract, as in ppl.nn.
#pragma once
#include "ppl/common/retcode.h"
#include "ppl/nn/common/device_context.h"
#include "ppl/nn/runtime/tensor.h"

namespace ppl { namespace nn {

class Runtime {
public:
    virtual ~Runtime() {}
    virtual uint32_t GetInputCount() const = 0;
    virtual Tensor* GetInputTensor(uint32_t idx) const = 0;
    virtual uint32_t GetOutputCount() const = 0;
    virtual Tensor* GetOutputTensor(uint32_t idx) const = 0;
    virtual uint32_t GetDeviceContextCount() const = 0;
    virtual DeviceContext* GetDeviceContext(uint32_t idx) const = 0;
    virtual ppl::common::RetCode Run() = 0;
};

}}  // namespace ppl::nn
