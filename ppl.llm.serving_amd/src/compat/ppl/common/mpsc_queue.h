// Stand-ins for ppl.common's MPSCQueue (intrusive), TypedMPSCQueue<T> and EventCount (contracts: SURVEY.md section
// 10; call sites src/utils/mpsc_request_scheduler.h:42-78, src/generator/llm_generator.h:107-144,
// src/generator/llm_generator.cc:346-359,812).
#pragma once
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>

namespace ppl { namespace common {

// Vyukov-style intrusive multi-producer single-consumer queue.
class MPSCQueue final {
public:
    struct Node {
        std::atomic<Node*> mpsc_next{nullptr};
        virtual ~Node() {}
    };

    MPSCQueue() : head_(&stub_), tail_(&stub_) {}

    void Push(Node* n) {  // any thread
        n->mpsc_next.store(nullptr, std::memory_order_relaxed);
        Node* prev = head_.exchange(n, std::memory_order_acq_rel);
        prev->mpsc_next.store(n, std::memory_order_release);
    }

    // single consumer.  Returns a node, or nullptr with *is_empty = true (really empty) / false (a producer is
    // between its two stores: spin and retry).
    Node* Pop(bool* is_empty) {
        Node* tail = tail_;
        Node* next = tail->mpsc_next.load(std::memory_order_acquire);
        if (tail == &stub_) {
            if (!next) {
                *is_empty = (head_.load(std::memory_order_acquire) == tail);
                return nullptr;
            }
            tail_ = next;
            tail = next;
            next = next->mpsc_next.load(std::memory_order_acquire);
        }
        if (next) {
            tail_ = next;
            *is_empty = false;
            return tail;
        }
        if (tail != head_.load(std::memory_order_acquire)) {
            *is_empty = false;  // producer mid-push
            return nullptr;
        }
        Push(&stub_);
        next = tail->mpsc_next.load(std::memory_order_acquire);
        if (next) {
            tail_ = next;
            *is_empty = false;
            return tail;
        }
        *is_empty = false;
        return nullptr;
    }

private:
    Node stub_;
    std::atomic<Node*> head_;
    Node* tail_;
};

template <typename T>
class TypedMPSCQueue final {
public:
    void Push(const T& v) {
        std::lock_guard<std::mutex> g(mu_);
        q_.push_back(v);
        size_.store((uint32_t)q_.size(), std::memory_order_relaxed);
    }
    bool Pop(T* out) {
        std::lock_guard<std::mutex> g(mu_);
        if (q_.empty()) return false;
        *out = q_.front();
        q_.pop_front();
        size_.store((uint32_t)q_.size(), std::memory_order_relaxed);
        return true;
    }
    uint32_t Size() const { return size_.load(std::memory_order_relaxed); }

private:
    std::mutex mu_;
    std::deque<T> q_;
    std::atomic<uint32_t> size_{0};
};

// folly-style event count: key = PrepareWait(); re-check the predicate; CancelWait() or CommitWait(key).
// NotifyOne() between PrepareWait and CommitWait makes CommitWait return immediately (no lost wake-up).
class EventCount final {
public:
    typedef uint64_t Key;
    Key PrepareWait() {
        std::lock_guard<std::mutex> g(mu_);
        ++waiters_;
        return epoch_;
    }
    void CancelWait() {
        std::lock_guard<std::mutex> g(mu_);
        --waiters_;
    }
    void CommitWait(Key key) {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return epoch_ != key; });
        --waiters_;
    }
    void NotifyOne() {
        std::lock_guard<std::mutex> g(mu_);
        ++epoch_;
        if (waiters_ > 0) cv_.notify_all();
    }

private:
    std::mutex mu_;
    std::condition_variable cv_;
    uint64_t epoch_ = 0;
    uint32_t waiters_ = 0;
};

}}  // namespace ppl::common
