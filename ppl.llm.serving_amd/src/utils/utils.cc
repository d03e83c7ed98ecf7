#include "utils.h"

#include <cstdlib>

namespace ppl { namespace llm { namespace utils {

void ParseTokens(const std::string& s, std::set<int>* tokens) {
    size_t pos = 0;
    while (pos <= s.size()) {
        size_t comma = s.find(',', pos);
        if (comma == std::string::npos) comma = s.size();
        if (comma > pos) tokens->insert(std::atoi(s.substr(pos, comma - pos).c_str()));
        pos = comma + 1;
    }
}

uint64_t HashCombine(uint64_t prev, const int32_t* vec, int32_t len) {
    uint64_t seed = (uint64_t)(int64_t)len;
    seed ^= prev + 0x9e3779b9ull + (seed << 6) + (seed >> 2);
    for (int32_t i = 0; i < len; ++i) {
        const uint32_t mixed = (uint32_t)vec[i] + 0x9e3779b9u;  // 32-bit wrap-around, then zero-extended
        seed ^= (uint64_t)mixed + (seed << 6) + (seed >> 2);
    }
    return seed;
}

}}}  // namespace ppl::llm::utils
