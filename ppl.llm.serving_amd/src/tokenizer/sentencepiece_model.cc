#include "sentencepiece_model.h"

#include <algorithm>
#include <cstring>
#include <fstream>
#include <queue>
#include <sstream>

namespace ppl { namespace llm {

namespace {

const char kSpaceSymbol[] = "\xE2\x96\x81";  // U+2581
const char kReplacement[] = "\xEF\xBF\xBD";  // U+FFFD

// ---- protobuf wire format (varint / 64-bit / length-delimited / 32-bit) ---------------------------------------------
struct Reader {
    const uint8_t* p;
    const uint8_t* end;
    bool ok = true;
    uint64_t Varint() {
        uint64_t v = 0;
        for (int shift = 0; p < end && shift < 64; shift += 7) {
            const uint8_t b = *p++;
            v |= (uint64_t)(b & 0x7f) << shift;
            if (!(b & 0x80)) return v;
        }
        ok = false;
        return 0;
    }
    // next field: number, wire type; for length-delimited fields [data, data + len) is returned in sub
    bool Next(uint32_t* num, uint32_t* wt, uint64_t* val, Reader* sub) {
        if (p >= end || !ok) return false;
        const uint64_t key = Varint();
        *num = (uint32_t)(key >> 3);
        *wt = (uint32_t)(key & 7);
        switch (*wt) {
            case 0: *val = Varint(); break;
            case 1: if (end - p < 8) { ok = false; return false; } memcpy(val, p, 8); p += 8; break;
            case 5: { if (end - p < 4) { ok = false; return false; } uint32_t v; memcpy(&v, p, 4); *val = v; p += 4; break; }
            case 2: {
                const uint64_t len = Varint();
                if (!ok || (uint64_t)(end - p) < len) { ok = false; return false; }
                sub->p = p; sub->end = p + len; sub->ok = true;
                p += len;
                break;
            }
            default: ok = false; return false;
        }
        return ok;
    }
};

// length in bytes of the UTF-8 character starting at s (1 for malformed input, as sentencepiece's OneCharLen + validity fallback)
size_t CharLen(const char* s, const char* end) {
    const uint8_t c = (uint8_t)*s;
    size_t n = c < 0x80 ? 1 : (c >> 5) == 0x6 ? 2 : (c >> 4) == 0xe ? 3 : (c >> 3) == 0x1e ? 4 : 1;
    if ((size_t)(end - s) < n) return 1;
    for (size_t i = 1; i < n; ++i)
        if (((uint8_t)s[i] & 0xc0) != 0x80) return 1;
    return n;
}

// true if [s, s + n) is one well-formed UTF-8 character (no overlongs, no surrogates, <= U+10FFFF)
bool ValidChar(const uint8_t* s, size_t avail, size_t* n) {
    const uint8_t c = s[0];
    uint32_t cp;
    size_t len;
    if (c < 0x80) { *n = 1; return true; }
    if ((c >> 5) == 0x6) { len = 2; cp = c & 0x1f; }
    else if ((c >> 4) == 0xe) { len = 3; cp = c & 0x0f; }
    else if ((c >> 3) == 0x1e) { len = 4; cp = c & 0x07; }
    else return false;
    if (avail < len) return false;
    for (size_t i = 1; i < len; ++i) {
        if ((s[i] & 0xc0) != 0x80) return false;
        cp = (cp << 6) | (s[i] & 0x3f);
    }
    if ((len == 2 && cp < 0x80) || (len == 3 && cp < 0x800) || (len == 4 && cp < 0x10000)) return false;
    if (cp > 0x10ffff || (cp >= 0xd800 && cp <= 0xdfff)) return false;
    *n = len;
    return true;
}

}  // namespace

bool SentencePieceModel::Load(const std::string& path, std::string* err) {
    std::ifstream f(path, std::ios::binary);
    if (!f) {
        if (err) *err = "cannot open " + path;
        return false;
    }
    std::stringstream ss;
    ss << f.rdbuf();
    return LoadFromBytes(ss.str(), err);
}

bool SentencePieceModel::LoadFromBytes(const std::string& bytes, std::string* err) {
    pieces_.clear();
    piece_to_id_.clear();
    std::fill(byte_to_id_, byte_to_id_ + 256, -1);
    Reader r{(const uint8_t*)bytes.data(), (const uint8_t*)bytes.data() + bytes.size()};
    uint32_t num, wt;
    uint64_t val;
    Reader sub{nullptr, nullptr};
    bool has_charsmap = false;
    while (r.Next(&num, &wt, &val, &sub)) {
        if (num == 1 && wt == 2) {            // repeated SentencePiece pieces
            Piece pc;
            Reader s2{nullptr, nullptr};
            uint32_t n2, w2;
            uint64_t v2;
            while (sub.Next(&n2, &w2, &v2, &s2)) {
                if (n2 == 1 && w2 == 2) pc.piece.assign((const char*)s2.p, s2.end - s2.p);
                else if (n2 == 2 && w2 == 5) { uint32_t u = (uint32_t)v2; memcpy(&pc.score, &u, 4); }
                else if (n2 == 3 && w2 == 0) pc.type = (int)v2;
            }
            pieces_.push_back(pc);
        } else if (num == 2 && wt == 2) {     // TrainerSpec
            Reader s2{nullptr, nullptr};
            uint32_t n2, w2;
            uint64_t v2;
            while (sub.Next(&n2, &w2, &v2, &s2)) {
                if (n2 == 3 && w2 == 0) model_type_ = (int)v2;
                else if (n2 == 35 && w2 == 0) byte_fallback_ = v2 != 0;
                else if (n2 == 40 && w2 == 0) unk_id_ = (int)(int32_t)v2;
                else if (n2 == 41 && w2 == 0) bos_id_ = (int)(int32_t)v2;
                else if (n2 == 42 && w2 == 0) eos_id_ = (int)(int32_t)v2;
                else if (n2 == 43 && w2 == 0) pad_id_ = (int)(int32_t)v2;
                else if (n2 == 44 && w2 == 2) unk_surface_.assign((const char*)s2.p, s2.end - s2.p);
            }
        } else if (num == 3 && wt == 2) {     // NormalizerSpec
            Reader s2{nullptr, nullptr};
            uint32_t n2, w2;
            uint64_t v2;
            while (sub.Next(&n2, &w2, &v2, &s2)) {
                if (n2 == 2 && w2 == 2) has_charsmap = s2.end > s2.p;
                else if (n2 == 3 && w2 == 0) add_dummy_prefix_ = v2 != 0;
                else if (n2 == 4 && w2 == 0) remove_extra_whitespaces_ = v2 != 0;
                else if (n2 == 5 && w2 == 0) escape_whitespaces_ = v2 != 0;
            }
        }
    }
    if (!r.ok || pieces_.empty()) {
        if (err) *err = "not a SentencePiece model (malformed ModelProto or no pieces)";
        return false;
    }
    if (has_charsmap) {
        if (err) *err = "the model carries a precompiled normalisation charsmap (e.g. nmt_nfkc); only identity-normalised models are supported";
        return false;
    }
    if (model_type_ != UNIGRAM && model_type_ != BPE) {
        if (err) *err = "unsupported SentencePiece model_type " + std::to_string(model_type_) + " (unigram and bpe only)";
        return false;
    }
    min_score_ = 0.f;
    bool first = true;
    for (size_t i = 0; i < pieces_.size(); ++i) {
        const Piece& pc = pieces_[i];
        if (pc.type == UNUSED) continue;
        piece_to_id_.emplace(pc.piece, (int)i);
        if (pc.type == BYTE && pc.piece.size() == 6 && pc.piece.compare(0, 3, "<0x") == 0 && pc.piece[5] == '>')
            byte_to_id_[strtol(pc.piece.substr(3, 2).c_str(), nullptr, 16)] = (int)i;
        if (pc.type == NORMAL || pc.type == USER_DEFINED) {
            if (first || pc.score < min_score_) min_score_ = pc.score;
            first = false;
            max_piece_bytes_ = std::max(max_piece_bytes_, pc.piece.size());
        }
    }
    if (byte_fallback_)
        for (int b = 0; b < 256; ++b)
            if (byte_to_id_[b] < 0) {
                if (err) *err = "byte_fallback model without all 256 byte pieces";
                return false;
            }
    return true;
}

int SentencePieceModel::PieceToId(const std::string& piece) const {
    auto it = piece_to_id_.find(piece);
    return it == piece_to_id_.end() ? unk_id_ : it->second;
}

// normalizer.cc (Normalizer::Normalize) with the identity character map: leading spaces dropped and runs of spaces merged when
// remove_extra_whitespaces, dummy prefix, ' ' -> U+2581, and -- as the library does it -- the trailing clean-up strips the SPACE
// SYMBOL from the normalised string, so a literal U+2581 at the end of the input goes too (and takes the dummy prefix with it
// when nothing else is left).
std::string SentencePieceModel::Normalize(const char* text, size_t len) const {
    size_t b = 0;
    if (remove_extra_whitespaces_)
        while (b < len && text[b] == ' ') ++b;
    std::string out;
    if (b == len) return out;
    out.reserve(len - b + 8);
    const std::string space = escape_whitespaces_ ? std::string(kSpaceSymbol) : std::string(" ");
    if (add_dummy_prefix_) out += space;
    bool is_prev_space = remove_extra_whitespaces_;
    for (size_t i = b; i < len; ++i) {
        const char c = text[i];
        if (c == ' ') {
            if (is_prev_space) continue;
            out += space;
            is_prev_space = remove_extra_whitespaces_;
        } else {
            out += c;
            is_prev_space = false;
        }
    }
    if (remove_extra_whitespaces_)
        while (out.size() >= space.size() && out.compare(out.size() - space.size(), space.size(), space) == 0)
            out.resize(out.size() - space.size());
    return out;
}

void SentencePieceModel::AppendPieceOrBytes(const std::string& sym, int id, std::vector<int>* ids) const {
    if (id != unk_id_ || !byte_fallback_) {
        ids->push_back(id);
        return;
    }
    for (unsigned char c : sym) ids->push_back(byte_to_id_[c]);   // bpe_model.cc / unigram_model.cc: unknown piece -> its bytes
}

// bpe_model.cc Model::Encode: symbols = characters; merge the best-scored adjacent pair until none is a piece
void SentencePieceModel::EncodeBpe(const std::string& norm, std::vector<int>* ids) const {
    struct Symbol { int prev, next; size_t pos, len; };
    std::vector<Symbol> sym;
    for (size_t i = 0; i < norm.size();) {
        const size_t n = CharLen(norm.data() + i, norm.data() + norm.size());
        sym.push_back(Symbol{(int)sym.size() - 1, (int)sym.size() + 1, i, n});
        i += n;
    }
    if (sym.empty()) return;
    sym.back().next = -1;
    struct Pair { int left, right; float score; size_t size; };
    auto cmp = [](const Pair& a, const Pair& b) { return a.score < b.score || (a.score == b.score && a.left > b.left); };
    std::priority_queue<Pair, std::vector<Pair>, decltype(cmp)> agenda(cmp);
    auto maybe_add = [&](int l, int r) {
        if (l < 0 || r < 0) return;
        const std::string piece = norm.substr(sym[l].pos, sym[l].len + sym[r].len);
        auto it = piece_to_id_.find(piece);
        if (it == piece_to_id_.end()) return;
        // merges run through NORMAL and USER_DEFINED pieces only: reserved ids (CONTROL "<s>", UNKNOWN, BYTE "<0x0A>") must never be
        // produced from literal text -- a tokenisation mismatch and a way to inject special tokens; UNUSED never merges
        const int ty = pieces_[it->second].type;
        if (ty != NORMAL && ty != USER_DEFINED) return;
        agenda.push(Pair{l, r, pieces_[it->second].score, piece.size()});
    };
    for (int i = 1; i < (int)sym.size(); ++i) maybe_add(i - 1, i);
    while (!agenda.empty()) {
        const Pair top = agenda.top();
        agenda.pop();
        Symbol& L = sym[top.left];
        Symbol& R = sym[top.right];
        if (L.len == 0 || R.len == 0 || L.len + R.len != top.size) continue;  // one side was merged away since
        L.len += R.len;
        L.next = R.next;
        if (R.next >= 0) sym[R.next].prev = top.left;
        R.len = 0;
        maybe_add(L.prev, top.left);
        maybe_add(top.left, L.next);
    }
    for (int i = 0; i >= 0; i = sym[i].next) {
        const std::string piece = norm.substr(sym[i].pos, sym[i].len);
        AppendPieceOrBytes(piece, PieceToId(piece), ids);
    }
}

// unigram_model.cc Model::EncodeOptimized: Viterbi, unknown characters cost min_score - 10
void SentencePieceModel::EncodeUnigram(const std::string& norm, std::vector<int>* ids) const {
    const size_t n = norm.size();
    if (n == 0) return;
    const float unk_score = min_score_ - 10.0f;
    struct Node { int id = -1; float best = 0.f; int starts_at = -1; };
    std::vector<Node> best(n + 1);
    size_t pos = 0;
    while (pos < n) {
        const float base = best[pos].best;
        const size_t mblen = std::min(CharLen(norm.data() + pos, norm.data() + n), n - pos);
        bool has_single = false;
        // pieces that start here, in increasing length (the trie's common-prefix search order)
        for (size_t len = 1; len <= max_piece_bytes_ && pos + len <= n; ++len) {
            auto it = piece_to_id_.find(norm.substr(pos, len));
            if (it == piece_to_id_.end()) continue;
            const Piece& pc = pieces_[it->second];
            if (pc.type != NORMAL && pc.type != USER_DEFINED) continue;
            Node& t = best[pos + len];
            const float length_bonus = pc.type == USER_DEFINED ? (float)len * 1.0f - 0.1f : 0.f;  // user-defined pieces always win
            const float cand = base + (pc.type == USER_DEFINED ? length_bonus : pc.score);
            if (t.starts_at == -1 || cand > t.best) { t.best = cand; t.starts_at = (int)pos; t.id = it->second; }
            if (!has_single && len == mblen) has_single = true;
        }
        if (!has_single) {
            Node& t = best[pos + mblen];
            const float cand = base + unk_score;
            if (t.starts_at == -1 || cand > t.best) { t.best = cand; t.starts_at = (int)pos; t.id = unk_id_; }
        }
        pos += mblen;  // (positions inside a character are unreachable)
    }
    std::vector<std::pair<int, int>> path;  // (start, id) back to front
    for (int end = (int)n; end > 0;) {
        const Node& nd = best[end];
        path.emplace_back(nd.starts_at, nd.id);
        end = nd.starts_at;
    }
    int end = (int)n;
    std::vector<std::pair<std::string, int>> seq;
    for (auto& pr : path) {
        seq.emplace_back(norm.substr(pr.first, end - pr.first), pr.second);
        end = pr.first;
    }
    bool prev_unk = false;
    for (auto it = seq.rbegin(); it != seq.rend(); ++it) {
        const bool unk = it->second == unk_id_;
        if (unk && prev_unk && !byte_fallback_) continue;   // a run of unknown characters is ONE <unk> (unigram_model.cc)
        AppendPieceOrBytes(it->first, it->second, ids);
        prev_unk = unk;
    }
}

void SentencePieceModel::Encode(const char* text, size_t len, std::vector<int>* ids) const {
    const std::string norm = Normalize(text, len);
    if (model_type_ == BPE) EncodeBpe(norm, ids);
    else EncodeUnigram(norm, ids);
}

// sentencepiece_processor.cc SentencePieceProcessor::Decode(ids)
void SentencePieceModel::Decode(const int* ids, size_t n, std::string* out) const {
    out->clear();
    std::string bytes;
    auto flush_bytes = [&]() {
        for (size_t i = 0; i < bytes.size();) {
            size_t cl = 0;
            if (ValidChar((const uint8_t*)bytes.data() + i, bytes.size() - i, &cl)) {
                out->append(bytes, i, cl);
                i += cl;
            } else {
                out->append(kReplacement);   // one U+FFFD per undecodable byte
                ++i;
            }
        }
        bytes.clear();
    };
    // the dummy prefix: ONE leading U+2581 is consumed from a piece while nothing has been emitted yet -- from the first such piece
    // only (add_dummy_prefix), or from every piece until text appears (remove_extra_whitespaces)
    bool is_bos_ws = true, bos_ws_seen = false;
    for (size_t i = 0; i < n; ++i) {
        const int id = ids[i];
        if (id < 0 || id >= (int)pieces_.size()) continue;
        const Piece& pc = pieces_[id];
        if (pc.type == BYTE) {
            bytes += (char)strtol(pc.piece.substr(3, 2).c_str(), nullptr, 16);
            continue;
        }
        flush_bytes();
        if (bos_ws_seen || !out->empty()) is_bos_ws = false;
        if (pc.type == CONTROL) continue;                      // <s>, </s>
        if (pc.type == UNKNOWN) { out->append(unk_surface_); continue; }
        size_t start = 0;
        if (is_bos_ws && (add_dummy_prefix_ || remove_extra_whitespaces_)) {
            if (pc.piece.compare(0, 3, kSpaceSymbol) == 0) {
                start = 3;
                bos_ws_seen = !remove_extra_whitespaces_;
            }
        }
        for (size_t k = start; k < pc.piece.size();) {
            if (pc.piece.compare(k, 3, kSpaceSymbol) == 0) { out->push_back(' '); k += 3; }
            else out->push_back(pc.piece[k++]);
        }
    }
    flush_bytes();
}

}}  // namespace ppl::llm
