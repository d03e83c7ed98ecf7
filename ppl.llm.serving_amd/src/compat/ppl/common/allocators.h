// Stand-ins for ppl.common's PageManager and the CompactAddrManager-backed range allocator behind
// utils::IndexManager (contracts: SURVEY.md section 10; call sites src/generator/llm_generator.cc:155-157,
// 487,516-554,373-376,418-425,563,775 and src/utils/index_manager.h:25-78).
#pragma once
#include <stdint.h>

#include <map>
#include <vector>

#include "retcode.h"

namespace ppl { namespace common {

// Pages of `page_size` tokens over a pool of max_tokens tokens; slot of (page, offset) = page*page_size + offset.
// Alloc APPENDS n page ids to `out` (the prefix-cache path appends after the cached pages,
// llm_generator.cc:534-546) and fails without a partial allocation.  Pages are handed out lowest-id-first from a
// LIFO free list seeded in ascending order, so a fresh manager yields 0, 1, 2, ...
class PageManager final {
public:
    void Init(uint64_t max_tokens, int64_t page_size) {
        free_.clear();
        if (page_size <= 0) return;
        const int64_t n = (int64_t)(max_tokens / (uint64_t)page_size);
        free_.reserve(n);
        for (int64_t i = n - 1; i >= 0; --i) free_.push_back(i);
    }
    RetCode Alloc(int64_t n, std::vector<int64_t>* out) {
        if (n < 0 || (int64_t)free_.size() < n) return RC_OUT_OF_MEMORY;
        for (int64_t i = 0; i < n; ++i) {
            out->push_back(free_.back());
            free_.pop_back();
        }
        return RC_SUCCESS;
    }
    void Free(const int64_t* ids, int64_t n) {
        for (int64_t i = 0; i < n; ++i) free_.push_back(ids[i]);
    }
    int64_t GetAvail() const { return (int64_t)free_.size(); }

private:
    std::vector<int64_t> free_;
};

// First-fit allocator of contiguous ranges [start, start+n) inside [0, max): lowest start wins, frees coalesce.
class RangeAllocator final {
public:
    void Init(uint64_t max) {
        free_.clear();
        if (max) free_[0] = max;
    }
    // returns the start or UINT64_MAX
    uint64_t Alloc(uint64_t n) {
        if (n == 0) return UINT64_MAX;
        for (auto it = free_.begin(); it != free_.end(); ++it) {
            if (it->second >= n) {
                const uint64_t start = it->first, len = it->second;
                free_.erase(it);
                if (len > n) free_[start + n] = len - n;
                return start;
            }
        }
        return UINT64_MAX;
    }
    void Free(uint64_t start, uint64_t n) {
        if (n == 0) return;
        auto next = free_.lower_bound(start);
        if (next != free_.begin()) {
            auto prev = std::prev(next);
            if (prev->first + prev->second == start) {
                start = prev->first;
                n += prev->second;
                free_.erase(prev);
            }
        }
        if (next != free_.end() && start + n == next->first) {
            n += next->second;
            free_.erase(next);
        }
        free_[start] = n;
    }

private:
    std::map<uint64_t, uint64_t> free_;  // start -> length
};

}}  // namespace ppl::common
