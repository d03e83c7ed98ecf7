// The ppl::nn::Tensor surface the reference drives (call sites src/engine/llm_engine.h:124-147,
// src/engine/llm_engine.cc:29-111,207-222, src/utils/utils.cc:110-163).  An abstract interface, as in ppl.nn; the HIP
// implementation is src/backends/hip_nn.
#pragma once
#include "ppl/common/retcode.h"
#include "ppl/nn/common/device_context.h"
#include "ppl/nn/runtime/tensor_shape.h"

namespace ppl { namespace nn {

class Tensor {
public:
    virtual ~Tensor() {}
    virtual const char* GetName() const = 0;
    virtual TensorShape* GetShape() const = 0;
    virtual DeviceContext* GetDeviceContext() const = 0;
    virtual void SetDeviceContext(DeviceContext*) = 0;
    virtual void SetBufferPtr(void*) = 0;           // the tensor aliases caller-owned device memory (the KV slab)
    virtual void* GetBufferPtr() const = 0;
    virtual ppl::common::RetCode ReallocBuffer() = 0;
    virtual void FreeBuffer() = 0;
    virtual ppl::common::RetCode CopyFromHostAsync(const void* src) = 0;  // src holds GetShape() elements
    virtual ppl::common::RetCode CopyFromHost(const void* src) = 0;
    virtual ppl::common::RetCode CopyToHost(void* dst) const = 0;
    virtual ppl::common::RetCode ConvertToHost(void* dst, const TensorShape& dst_desc) const = 0;
};

}}  // namespace ppl::nn
