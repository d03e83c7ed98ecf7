// Multi-producer single-consumer request queue with a one-element stash: a head request that the admission check
// rejects is kept and retried first on the next call (reference src/utils/mpsc_request_scheduler.h:30-106).
#pragma once
#include <atomic>
#include <functional>
#include <type_traits>

#include "ppl/common/mpsc_queue.h"

namespace ppl { namespace llm { namespace utils {

template <typename ReqType>
class MPSCRequestScheduler final {
    static_assert(std::is_base_of<ppl::common::MPSCQueue::Node, ReqType>::value, "ReqType must derive from MPSCQueue::Node");

public:
    MPSCRequestScheduler() {}
    ~MPSCRequestScheduler() {
        delete stash_;
        bool empty = false;
        while (!empty) {
            auto* n = queue_.Pop(&empty);
            if (n) delete static_cast<ReqType*>(n);
        }
    }
    MPSCRequestScheduler(const MPSCRequestScheduler&) = delete;
    void operator=(const MPSCRequestScheduler&) = delete;

    // true if the queue MAY have been empty before this push (the caller then wakes the consumer)
    bool PushRequest(ReqType* req) {
        queue_.Push(req);
        return pending_.fetch_add(1, std::memory_order_acq_rel) == 0;
    }

    // consumer only.  Returns the next request if `admit` accepts it, else nullptr (and remembers the request).
    ReqType* TryPopRequest(const std::function<bool(const ReqType&)>& admit) {
        ReqType* req = stash_;
        if (!req) {
            bool empty = true;
            ppl::common::MPSCQueue::Node* n = nullptr;
            do { n = queue_.Pop(&empty); } while (!n && !empty);
            if (!n) return nullptr;
            req = static_cast<ReqType*>(n);
        }
        if (!admit(*req)) {
            stash_ = req;
            return nullptr;
        }
        stash_ = nullptr;
        pending_.fetch_sub(1, std::memory_order_acq_rel);
        return req;
    }

    uint32_t GetPendingSize() const { return pending_.load(std::memory_order_relaxed); }  // approximate

private:
    ppl::common::MPSCQueue queue_;
    std::atomic<uint32_t> pending_{0};
    ReqType* stash_ = nullptr;
};

}}}  // namespace ppl::llm::utils
