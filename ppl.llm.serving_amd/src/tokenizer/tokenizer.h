// The tokenizer interface of the text path (reference src/tokenizer/tokenizer.h:27-35).  The hot path is token-in/token-out
// (src/generator/llm_generator.cc:790-801); a tokenizer is only needed for text requests and for detokenising responses.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace ppl { namespace llm {

class Tokenizer {
public:
    virtual ~Tokenizer() {}
    virtual void Encode(const char* prompt, uint32_t len, std::vector<int>* token_ids) const = 0;
    virtual void Decode(int* token_ids, uint32_t len, std::string* output) const = 0;
    virtual int GetBosId() const = 0;
    virtual int GetEosId() const = 0;
};

}}  // namespace ppl::llm
