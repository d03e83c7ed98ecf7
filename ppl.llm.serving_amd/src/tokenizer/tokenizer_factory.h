// Tokenizer construction as in the reference (src/tokenizer/tokenizer_factory.h:37-86): a tokenizer implementation selected by
// --tokenizer-type ("sentencepiece"; the HuggingFace json tokenizer is a compile-time option there and not provided here) wrapped
// by the model family's policy (--model-type "llama": BOS in front of every prompt, models/llama/llama_tokenizer.h:35-38).
#pragma once
#include <memory>
#include <string>

#include "ppl/common/log.h"
#include "sentencepiece_model.h"
#include "tokenizer.h"

namespace ppl { namespace llm {

// reference src/tokenizer/tokenizer_impl_sp.h:31-74
class SentencePieceTokenizer final : public Tokenizer {
public:
    bool Init(const std::string& path) {
        std::string err;
        if (!sp_.Load(path, &err)) {
            LOG(ERROR) << "sentencepiece tokenizer init failed: " << err;
            return false;
        }
        LOG(INFO) << "VOCAB_SIZE: " << sp_.GetPieceSize() << "; BOS ID: " << sp_.bos_id() << "; EOS ID: " << sp_.eos_id()
                  << "; PAD ID: " << sp_.pad_id();
        return true;
    }
    void Encode(const char* prompt, uint32_t len, std::vector<int>* token_ids) const override { sp_.Encode(prompt, len, token_ids); }
    // a single piece that starts a word decodes without its space (the dummy-prefix rule): put it back, so that streamed pieces
    // concatenate to the text (tokenizer_impl_sp.h:53-59)
    void Decode(int* token_ids, uint32_t len, std::string* output) const override {
        sp_.Decode(token_ids, len, output);
        if (len == 1 && token_ids[0] >= 0 && token_ids[0] < sp_.GetPieceSize() && sp_.IdToPiece(token_ids[0]).compare(0, 3, "\xe2\x96\x81") == 0 &&
            !output->empty() && output->at(0) != ' ')
            output->insert(0, " ");
    }
    int GetBosId() const override { return sp_.bos_id(); }
    int GetEosId() const override { return sp_.eos_id(); }

private:
    SentencePieceModel sp_;
};

// The model family's policy around a tokenizer implementation (reference src/tokenizer/models/*/ *_tokenizer.h): llama, internlm and
// llama3 put BOS in front of every prompt (models/llama/llama_tokenizer.h:35-38, internlm_tokenizer.h, llama3_tokenizer.h),
// baichuan encodes the prompt as it is (models/baichuan/baichuan_tokenizer.h).
class LlamaTokenizer final : public Tokenizer {
public:
    explicit LlamaTokenizer(Tokenizer* impl, bool prepend_bos = true) : impl_(impl), prepend_bos_(prepend_bos) {}
    void Encode(const char* prompt, uint32_t len, std::vector<int>* token_ids) const override {
        impl_->Encode(prompt, len, token_ids);
        if (prepend_bos_) token_ids->insert(token_ids->begin(), impl_->GetBosId());
    }
    void Decode(int* token_ids, uint32_t len, std::string* output) const override { impl_->Decode(token_ids, len, output); }
    int GetBosId() const override { return impl_->GetBosId(); }
    int GetEosId() const override { return impl_->GetEosId(); }

private:
    std::unique_ptr<Tokenizer> impl_;
    bool prepend_bos_;
};

class TokenizerFactory final {
public:
    static Tokenizer* Create(const std::string& model_type, const std::string& tokenizer_type, const std::string& tokenizer_path,
                             const std::string& /*tokenizer_config_path*/) {
        if (tokenizer_type != "sentencepiece") {
            LOG(ERROR) << "not supported tokenizer: " << tokenizer_type;
            return nullptr;
        }
        std::unique_ptr<SentencePieceTokenizer> impl(new SentencePieceTokenizer());
        if (!impl->Init(tokenizer_path)) return nullptr;
        if (model_type == "llama" || model_type == "internlm" || model_type == "llama3") return new LlamaTokenizer(impl.release(), true);
        if (model_type == "baichuan") return new LlamaTokenizer(impl.release(), false);
        LOG(ERROR) << "not supported model: " << model_type;
        return nullptr;
    }
};

}}  // namespace ppl::llm
