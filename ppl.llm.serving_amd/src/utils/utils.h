// Host utilities of the hot path (reference src/utils/utils.h:37-68, utils.cc:78-94).
#pragma once
#include <chrono>
#include <set>
#include <string>
#include <vector>

#include "ppl/common/log.h"
#include "ppl/common/retcode.h"
#include "ppl/common/threadpool.h"

namespace ppl { namespace llm { namespace utils {

// "1,2,3" -> {1,2,3}; empty fields are skipped (reference ParseTokens, src/utils/utils.cc:66-76)
void ParseTokens(const std::string& tokens_str, std::set<int>* tokens);

// Runs func(ithr, args...) on every thread of the pool (one thread per tensor-parallel rank) and joins.
// Deviation from the reference (src/utils/utils.h:45-49): the reference returns thr_rc[0] from inside
// the loop, ignoring failures on ranks >= 1; here the first failing rank's code is returned.
template <class F, typename... Args>
ppl::common::RetCode ParallelExecute(F&& func, ppl::common::StaticThreadPool* pool, Args&&... args) {
    const uint32_t n = pool->GetNumThreads();
    std::vector<ppl::common::RetCode> rc(n, ppl::common::RC_SUCCESS);
    pool->Run([&](uint32_t, uint32_t ithr) { rc[ithr] = func(ithr, args...); });
    for (uint32_t i = 0; i < n; ++i) {
        if (rc[i] != ppl::common::RC_SUCCESS) {
            LOG(ERROR) << "ParallelExecute task[" << i << "] failed";
            return rc[i];
        }
    }
    return ppl::common::RC_SUCCESS;
}

// RAII wall-clock timer writing microseconds into *res (reference TimingGuard, src/utils/utils.h:54-68)
class TimingGuard final {
public:
    explicit TimingGuard(uint64_t* res) : out_(res), t0_(std::chrono::high_resolution_clock::now()) {}
    ~TimingGuard() {
        *out_ = (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::high_resolution_clock::now() - t0_).count();
    }

private:
    uint64_t* out_;
    std::chrono::time_point<std::chrono::high_resolution_clock> t0_;
};

// boost-style hash chaining of one KV page of token ids (reference src/utils/utils.cc:87-94).  NOTE the mixed
// width: `vec[i] + 0x9e3779b9` is evaluated in 32-bit unsigned arithmetic and then widened, while
// `prev + 0x9e3779b9` is 64-bit -- reproduced bit for bit (golden values: tests/golden/host_logic.json).
uint64_t HashCombine(uint64_t prev, const int32_t* vec, int32_t len);

}}}  // namespace ppl::llm::utils
