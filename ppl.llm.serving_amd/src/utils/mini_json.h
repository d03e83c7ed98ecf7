// A small JSON reader (the reference uses rapidjson, which is not available here): objects, arrays, strings with the
// common escapes, numbers, true/false/null.  Enough for params.json and the test scenarios.
#pragma once
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace ppl { namespace llm { namespace utils {

struct JsonValue {
    enum Type { NUL, BOOL, NUMBER, STRING, ARRAY, OBJECT } type = NUL;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<JsonValue> arr;
    std::map<std::string, JsonValue> obj;

    const JsonValue* Find(const std::string& key) const {
        if (type != OBJECT) return nullptr;
        auto it = obj.find(key);
        return it == obj.end() ? nullptr : &it->second;
    }
    int64_t AsInt() const { return (int64_t)num; }
    int64_t GetInt(const std::string& key, int64_t dflt) const {
        const JsonValue* v = Find(key);
        return v && v->type == NUMBER ? (int64_t)v->num : dflt;
    }
    double GetNum(const std::string& key, double dflt) const {
        const JsonValue* v = Find(key);
        return v && v->type == NUMBER ? v->num : dflt;
    }
    std::string GetString(const std::string& key, const std::string& dflt) const {
        const JsonValue* v = Find(key);
        return v && v->type == STRING ? v->str : dflt;
    }
    bool GetBool(const std::string& key, bool dflt) const {
        const JsonValue* v = Find(key);
        if (!v) return dflt;
        return v->type == BOOL ? v->b : (v->type == NUMBER ? v->num != 0 : dflt);
    }
};

class JsonParser final {
public:
    explicit JsonParser(const std::string& s) : s_(s) {}
    bool Parse(JsonValue* out) {
        pos_ = 0;
        if (!Value(out)) return false;
        Skip();
        return pos_ == s_.size();
    }

private:
    bool At(char c) const { return pos_ < s_.size() && s_[pos_] == c; }
    void Skip() {
        while (pos_ < s_.size() && (s_[pos_] == ' ' || s_[pos_] == '\n' || s_[pos_] == '\t' || s_[pos_] == '\r')) ++pos_;
    }
    bool Lit(const char* w) {
        const size_t n = strlen(w);
        if (s_.compare(pos_, n, w) != 0) return false;
        pos_ += n;
        return true;
    }
    bool String(std::string* out) {
        if (!At('"')) return false;
        ++pos_;
        out->clear();
        while (pos_ < s_.size() && s_[pos_] != '"') {
            char c = s_[pos_++];
            if (c == '\\' && pos_ < s_.size()) {
                const char e = s_[pos_++];
                switch (e) {
                    case 'n': c = '\n'; break;
                    case 't': c = '\t'; break;
                    case 'r': c = '\r'; break;
                    case 'b': c = '\b'; break;
                    case 'f': c = '\f'; break;
                    case 'u': {  // BMP code point -> UTF-8
                        if (pos_ + 4 > s_.size()) return false;
                        const unsigned cp = (unsigned)strtoul(s_.substr(pos_, 4).c_str(), nullptr, 16);
                        pos_ += 4;
                        if (cp < 0x80) {
                            out->push_back((char)cp);
                        } else if (cp < 0x800) {
                            out->push_back((char)(0xC0 | (cp >> 6)));
                            out->push_back((char)(0x80 | (cp & 0x3F)));
                        } else {
                            out->push_back((char)(0xE0 | (cp >> 12)));
                            out->push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
                            out->push_back((char)(0x80 | (cp & 0x3F)));
                        }
                        continue;
                    }
                    default: c = e;
                }
            }
            out->push_back(c);
        }
        if (pos_ >= s_.size()) return false;
        ++pos_;
        return true;
    }
    bool Value(JsonValue* v) {
        Skip();
        if (pos_ >= s_.size()) return false;
        const char c = s_[pos_];
        if (c == '{') {
            v->type = JsonValue::OBJECT;
            ++pos_;
            Skip();
            if (At('}')) { ++pos_; return true; }
            while (true) {
                Skip();
                std::string key;
                if (!String(&key)) return false;
                Skip();
                if (!At(':')) return false;
                ++pos_;
                if (!Value(&v->obj[key])) return false;
                Skip();
                if (At(',')) { ++pos_; continue; }
                if (At('}')) { ++pos_; return true; }
                return false;
            }
        }
        if (c == '[') {
            v->type = JsonValue::ARRAY;
            ++pos_;
            Skip();
            if (At(']')) { ++pos_; return true; }
            while (true) {
                v->arr.emplace_back();
                if (!Value(&v->arr.back())) return false;
                Skip();
                if (At(',')) { ++pos_; continue; }
                if (At(']')) { ++pos_; return true; }
                return false;
            }
        }
        if (c == '"') { v->type = JsonValue::STRING; return String(&v->str); }
        if (c == 't') { v->type = JsonValue::BOOL; v->b = true; return Lit("true"); }
        if (c == 'f') { v->type = JsonValue::BOOL; v->b = false; return Lit("false"); }
        if (c == 'n') { v->type = JsonValue::NUL; return Lit("null"); }
        char* end = nullptr;
        v->num = strtod(s_.c_str() + pos_, &end);
        if (end == s_.c_str() + pos_) return false;
        v->type = JsonValue::NUMBER;
        pos_ = end - s_.c_str();
        return true;
    }

    const std::string& s_;
    size_t pos_ = 0;
};

}}}  // namespace ppl::llm::utils
