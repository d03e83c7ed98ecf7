// A reader and encoder / decoder for SentencePiece model files (the `tokenizer.model` of LLaMA checkpoints), written against the
// published format and algorithms of google/sentencepiece -- the library the reference links (src/tokenizer/tokenizer_impl_sp.h:31-74,
// fetched from github.com/OpenPPL/sentencepiece branch `ppl`, cmake/deps.cmake:124-134; not in the tree, not in this image as C++).
// What is implemented is what LLaMA-family models use:
//   * the ModelProto wire format (sentencepiece_model.proto: pieces {piece, score, type}, trainer_spec {model_type, byte_fallback,
//     unk/bos/eos/pad ids, unk_surface}, normalizer_spec {add_dummy_prefix, remove_extra_whitespaces, escape_whitespaces});
//     a precompiled normalisation charsmap (NFKC ...) is NOT applied -- LLaMA models are trained with the identity rule, and Load()
//     refuses a model that carries one;
//   * BPE encoding (bpe_model.cc: repeatedly merge the adjacent pair whose concatenation is the best-scored piece, leftmost on
//     ties) and unigram encoding (unigram_model.cc: Viterbi over piece scores, unknown characters at min_score - 10), both with
//     byte fallback (<0xXX> pieces) for characters outside the vocabulary;
//   * decoding (sentencepiece_processor.cc): control pieces vanish, byte pieces are joined and checked as UTF-8 (each invalid
//     byte becomes U+FFFD), U+2581 becomes a space, the dummy prefix is dropped from the first piece.
// Pinned against the `sentencepiece` Python module (same library, 0.2.x) on models trained in the build container:
// tests/test_tokenizer.py, fixtures tests/golden/spm_*.model + spm_cases.json.
#pragma once
#include <stdint.h>

#include <string>
#include <unordered_map>
#include <vector>

namespace ppl { namespace llm {

class SentencePieceModel final {
public:
    enum PieceType { NORMAL = 1, UNKNOWN = 2, CONTROL = 3, USER_DEFINED = 4, UNUSED = 5, BYTE = 6 };
    enum ModelType { UNIGRAM = 1, BPE = 2, WORD = 3, CHAR = 4 };

    bool Load(const std::string& path, std::string* err);
    bool LoadFromBytes(const std::string& bytes, std::string* err);

    void Encode(const char* text, size_t len, std::vector<int>* ids) const;
    void Decode(const int* ids, size_t n, std::string* out) const;

    int GetPieceSize() const { return (int)pieces_.size(); }
    const std::string& IdToPiece(int id) const { return pieces_[id].piece; }
    int PieceToId(const std::string& piece) const;
    int bos_id() const { return bos_id_; }
    int eos_id() const { return eos_id_; }
    int unk_id() const { return unk_id_; }
    int pad_id() const { return pad_id_; }

private:
    struct Piece {
        std::string piece;
        float score = 0.f;
        int type = NORMAL;
    };
    std::string Normalize(const char* text, size_t len) const;
    void EncodeBpe(const std::string& norm, std::vector<int>* ids) const;
    void EncodeUnigram(const std::string& norm, std::vector<int>* ids) const;
    void AppendPieceOrBytes(const std::string& sym, int id, std::vector<int>* ids) const;

    std::vector<Piece> pieces_;
    std::unordered_map<std::string, int> piece_to_id_;  // NORMAL / USER_DEFINED / UNKNOWN / CONTROL / BYTE (not UNUSED)
    int byte_to_id_[256];
    int model_type_ = UNIGRAM;
    bool byte_fallback_ = false;
    bool add_dummy_prefix_ = true, remove_extra_whitespaces_ = true, escape_whitespaces_ = true;
    int unk_id_ = 0, bos_id_ = 1, eos_id_ = 2, pad_id_ = -1;
    std::string unk_surface_ = " \xE2\x81\x87 ";
    float min_score_ = 0.f;
    size_t max_piece_bytes_ = 0;
};

}}  // namespace ppl::llm
