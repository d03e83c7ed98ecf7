// Stand-in for ppl.common's StaticThreadPool + Barrier (contracts inferred from the reference's call sites:
// src/utils/utils.h:39-44, src/backends/cuda/resource_manager.cc:410-418, src/generator/llm_generator.cc:176,622,
// 696,738,784): Init(n) spawns n persistent threads; Run(f) invokes f(nthr, ithr) once on every thread and blocks;
// RunAsync(f) returns immediately; Wait() joins the outstanding async batch (a no-op when nothing is pending).
#pragma once
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "retcode.h"

namespace ppl { namespace common {

class ThreadTask {};  // only named by utils::DummyTaskDeleter in the reference

class Barrier final {
public:
    void Reset(uint32_t n) {
        std::lock_guard<std::mutex> g(mu_);
        total_ = n;
        arrived_ = 0;
        ++generation_;
    }
    void Wait() {
        std::unique_lock<std::mutex> lk(mu_);
        const uint64_t gen = generation_;
        if (++arrived_ >= total_) {
            arrived_ = 0;
            ++generation_;
            cv_.notify_all();
            return;
        }
        cv_.wait(lk, [&] { return generation_ != gen; });
    }

private:
    std::mutex mu_;
    std::condition_variable cv_;
    uint32_t total_ = 0, arrived_ = 0;
    uint64_t generation_ = 0;
};

class StaticThreadPool final {
public:
    typedef std::function<void(uint32_t nthr, uint32_t ithr)> Func;

    StaticThreadPool() {}
    ~StaticThreadPool() { Destroy(); }

    RetCode Init(uint32_t n) {
        Destroy();
        if (n == 0) return RC_INVALID_VALUE;
        stop_ = false;
        epoch_ = 0;
        pending_ = 0;
        threads_.reserve(n);
        for (uint32_t i = 0; i < n; ++i) threads_.emplace_back([this, n, i] { Loop(n, i); });
        return RC_SUCCESS;
    }
    uint32_t GetNumThreads() const { return (uint32_t)threads_.size(); }

    void RunAsync(const Func& f) {
        Wait();
        std::lock_guard<std::mutex> g(mu_);
        func_ = f;
        pending_ = (uint32_t)threads_.size();
        ++epoch_;
        cv_work_.notify_all();
    }
    void Wait() {
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [&] { return pending_ == 0; });
    }
    void Run(const Func& f) {
        RunAsync(f);
        Wait();
    }

private:
    void Loop(uint32_t n, uint32_t i) {
        uint64_t seen = 0;
        while (true) {
            Func f;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_work_.wait(lk, [&] { return stop_ || epoch_ != seen; });
                if (stop_) return;
                seen = epoch_;
                f = func_;
            }
            f(n, i);
            {
                std::lock_guard<std::mutex> g(mu_);
                if (--pending_ == 0) cv_done_.notify_all();
            }
        }
    }
    void Destroy() {
        if (threads_.empty()) return;
        Wait();
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
            cv_work_.notify_all();
        }
        for (auto& t : threads_) t.join();
        threads_.clear();
    }

    std::vector<std::thread> threads_;
    std::mutex mu_;
    std::condition_variable cv_work_, cv_done_;
    Func func_;
    uint64_t epoch_ = 0;
    uint32_t pending_ = 0;
    bool stop_ = false;
};

}}  // namespace ppl::common
