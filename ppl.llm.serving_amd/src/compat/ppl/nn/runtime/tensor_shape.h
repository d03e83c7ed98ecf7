// The ppl::nn::TensorShape surface the reference uses (src/engine/llm_engine.cc:31-169, src/utils/utils.cc:110-163).
#pragma once
#include <stdint.h>

#include <initializer_list>
#include <vector>

#include "ppl/common/types.h"

namespace ppl { namespace nn {

class TensorShape final {
public:
    void Reshape(std::initializer_list<int64_t> dims) { dims_.assign(dims.begin(), dims.end()); scalar_ = false; }
    void Reshape(const std::vector<int64_t>& dims) { dims_ = dims; scalar_ = false; }
    void Reshape(const int64_t* dims, uint32_t n) { dims_.assign(dims, dims + n); scalar_ = false; }
    void ReshapeAsScalar() { dims_.clear(); scalar_ = true; }
    bool IsScalar() const { return scalar_; }
    uint32_t GetDimCount() const { return (uint32_t)dims_.size(); }
    uint32_t GetRealDimCount() const { return (uint32_t)dims_.size(); }
    int64_t GetDim(uint32_t i) const { return dims_[i]; }
    const int64_t* GetDims() const { return dims_.data(); }
    ppl::common::datatype_t GetDataType() const { return dtype_; }
    void SetDataType(ppl::common::datatype_t t) { dtype_ = t; }
    ppl::common::dataformat_t GetDataFormat() const { return format_; }
    void SetDataFormat(ppl::common::dataformat_t f) { format_ = f; }
    uint64_t CalcElementsIncludingPadding() const {
        uint64_t n = 1;
        for (int64_t d : dims_) n *= (uint64_t)d;
        return (dims_.empty() && !scalar_) ? 0 : n;
    }
    uint64_t CalcBytesIncludingPadding() const { return CalcElementsIncludingPadding() * ppl::common::GetSizeOfDataType(dtype_); }
    uint64_t CalcElementsExcludingPadding() const { return CalcElementsIncludingPadding(); }
    uint64_t CalcBytesExcludingPadding() const { return CalcBytesIncludingPadding(); }

private:
    std::vector<int64_t> dims_;
    bool scalar_ = false;
    ppl::common::datatype_t dtype_ = ppl::common::DATATYPE_UNKNOWN;
    ppl::common::dataformat_t format_ = ppl::common::DATAFORMAT_NDARRAY;
};

}}  // namespace ppl::nn
