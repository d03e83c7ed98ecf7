// Prefix cache: hash of a full KV page (chained over the prompt) -> page id, with reference counts; pages whose count
// drops to zero enter an LRU list and are evicted oldest-released-first when the page pool runs short
// (reference src/utils/prefix_cache_manager.h:14-186; golden behaviour: tests/golden/host_logic.json).
#pragma once
#include <stdint.h>

#include <list>
#include <unordered_map>
#include <vector>

#include "ppl/common/log.h"

namespace ppl { namespace llm { namespace utils {

class PrefixCacheManager final {
public:
    // page id of `hash_val`, or -1
    int64_t Find(uint64_t hash_val) const {
        auto it = entries_.find(hash_val);
        return it == entries_.end() ? -1 : it->second.page_id;
    }

    // a new entry starts with one reference (its inserting request); an existing hash is left untouched
    void Insert(uint64_t hash_val, int64_t page_id) {
        Entry e;
        e.page_id = page_id;
        e.ref_count = 1;
        e.in_lru = false;
        entries_.insert({hash_val, e});
    }

    void IncRefCount(const uint64_t* hash_list, int64_t nums) {
        for (int64_t i = 0; i < nums; ++i) {
            auto it = entries_.find(hash_list[i]);
            if (it == entries_.end()) {
                LOG(WARNING) << "hash [" << hash_list[i] << "] not found in prefix map";
                break;
            }
            ++it->second.ref_count;
            if (it->second.in_lru) {  // in use again: no longer evictable
                lru_.erase(it->second.lru_pos);
                it->second.in_lru = false;
            }
        }
    }

    void DecRefCount(const uint64_t* hash_list, int64_t nums) {
        for (int64_t i = 0; i < nums; ++i) {
            auto it = entries_.find(hash_list[i]);
            if (it == entries_.end()) {
                LOG(WARNING) << "hash [" << hash_list[i] << "] not found in prefix map";
                break;
            }
            if (--it->second.ref_count == 0 && !it->second.in_lru) {
                lru_.push_front(hash_list[i]);  // most recently released at the front
                it->second.lru_pos = lru_.begin();
                it->second.in_lru = true;
            }
        }
    }

    // evicts up to `nums` unreferenced pages, least recently released first; appends their page ids
    void Evict(int64_t nums, std::vector<int64_t>* page_list) {
        while (nums-- > 0 && !lru_.empty()) {
            const uint64_t h = lru_.back();
            lru_.pop_back();
            auto it = entries_.find(h);
            page_list->push_back(it->second.page_id);
            entries_.erase(it);
        }
    }

    int32_t Size() const { return (int32_t)entries_.size(); }

    void Reset() {
        entries_.clear();
        lru_.clear();
    }

private:
    struct Entry {
        int64_t page_id;
        int32_t ref_count;
        bool in_lru;
        std::list<uint64_t>::iterator lru_pos;
    };
    std::unordered_map<uint64_t, Entry> entries_;
    std::list<uint64_t> lru_;
};

}}}  // namespace ppl::llm::utils
