// Contiguous KV-slot ranges for cache_mode 0 and batch slots for the penalty count map (reference
// src/utils/index_manager.h:25-78; the CompactAddrManager behind it is external -- this build's allocator is
// first-fit, lowest start, coalescing frees: ppl/common/allocators.h).
#pragma once
#include <stdint.h>

#include "ppl/common/allocators.h"

namespace ppl { namespace llm { namespace utils {

class IndexManager final {
public:
    void Init(uint64_t max_index) {
        avail_ = max_index;
        ranges_.Init(max_index);
    }
    int64_t GetAvailableBlockNum() const { return (int64_t)avail_; }
    // start of a range of `nr` slots, or INT64_MAX
    int64_t Alloc(uint64_t nr) {
        const uint64_t start = ranges_.Alloc(nr);
        if (start == UINT64_MAX) return INT64_MAX;
        avail_ -= nr;
        return (int64_t)start;
    }
    void Free(uint64_t start, uint64_t nr) {
        ranges_.Free(start, nr);
        avail_ += nr;
    }

private:
    uint64_t avail_ = 0;
    ppl::common::RangeAllocator ranges_;
};

}}}  // namespace ppl::llm::utils
