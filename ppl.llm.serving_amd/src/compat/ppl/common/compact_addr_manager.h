// Stand-in for ppl.common's CompactAddrManager, the range allocator behind the reference's utils::IndexManager
// (src/utils/index_manager.h:25-78): contiguous ranges out of an address space that grows through
// VMAllocator::Extend(needed) (-> bytes granted, or 0), first fit, frees coalesce, Alloc -> start or UINTPTR_MAX.
// Same placement policy as RangeAllocator (allocators.h), which the repo's own IndexManager uses: lowest start wins.
#pragma once
#include <stdint.h>

#include <map>

namespace ppl { namespace common {

class CompactAddrManager final {
public:
    class VMAllocator {
    public:
        virtual ~VMAllocator() {}
        virtual uintptr_t GetReservedBase() const = 0;
        virtual uint64_t GetAllocatedSize() const = 0;
        virtual uint64_t Extend(uint64_t needed) = 0;
    };

    explicit CompactAddrManager(VMAllocator* vmr) : vmr_(vmr) {}

    uintptr_t Alloc(uint64_t n) {
        if (n == 0) return UINTPTR_MAX;
        for (auto it = free_.begin(); it != free_.end(); ++it) {  // first fit, lowest start first
            if (it->second >= n) {
                const uint64_t start = it->first, len = it->second;
                free_.erase(it);
                if (len > n) free_[start + n] = len - n;
                return vmr_->GetReservedBase() + start;
            }
        }
        // grow at the end; a free block that touches the end is extended instead of skipped
        uint64_t end = vmr_->GetAllocatedSize(), tail = 0;
        if (!free_.empty()) {
            auto last = std::prev(free_.end());
            if (last->first + last->second == end) tail = last->second;
        }
        if (vmr_->Extend(n - tail) != n - tail) return UINTPTR_MAX;
        uint64_t start = end;
        if (tail) {
            start = end - tail;
            free_.erase(std::prev(free_.end()));
        }
        return vmr_->GetReservedBase() + start;
    }

    void Free(uintptr_t addr, uint64_t n) {
        uint64_t start = addr - vmr_->GetReservedBase();
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
    VMAllocator* vmr_;
    std::map<uint64_t, uint64_t> free_;  // start -> length
};

}}  // namespace ppl::common
