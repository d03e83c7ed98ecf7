// Test driver for ppl.llm.serving_amd/src/tokenizer (TEST INFRASTRUCTURE).  Line protocol on stdin, hex so that any byte survives:
//   E <hex utf-8 text>   -> the ids of SentencePieceModel::Encode, space separated
//   D <id> <id> ...      -> hex of SentencePieceModel::Decode
//   L <hex text>         -> ids of LlamaTokenizer::Encode (BOS first; reference models/llama/llama_tokenizer.h:35-38)
//   B <hex text>         -> ids of the "baichuan" policy (no BOS; models/baichuan/baichuan_tokenizer.h)
//   T <id>               -> hex of the one-token decode the generator uses (leading-space rule of tokenizer_impl_sp.h:53-59)
#include <iostream>
#include <sstream>

#include "tokenizer/tokenizer_factory.h"

using namespace ppl::llm;

static std::string FromHex(const std::string& h) {
    std::string s;
    for (size_t i = 0; i + 1 < h.size(); i += 2) s.push_back((char)strtol(h.substr(i, 2).c_str(), nullptr, 16));
    return s;
}
static std::string ToHex(const std::string& s) {
    static const char* d = "0123456789abcdef";
    std::string h;
    for (unsigned char c : s) { h.push_back(d[c >> 4]); h.push_back(d[c & 15]); }
    return h;
}

int main(int argc, char** argv) {
    if (argc != 2) { std::cerr << "usage: tokenizer_tool <tokenizer.model>\n"; return 2; }
    SentencePieceModel sp;
    std::string err;
    if (!sp.Load(argv[1], &err)) { std::cout << "ERROR " << err << std::endl; return 1; }
    std::unique_ptr<Tokenizer> tok(TokenizerFactory::Create("llama", "sentencepiece", argv[1], ""));
    std::unique_ptr<Tokenizer> tok_b(TokenizerFactory::Create("baichuan", "sentencepiece", argv[1], ""));
    if (!tok || !tok_b) { std::cout << "ERROR factory" << std::endl; return 1; }
    std::cout << "OK " << sp.GetPieceSize() << " " << sp.bos_id() << " " << sp.eos_id() << " " << sp.unk_id() << std::endl;
    std::string line;
    while (std::getline(std::cin, line)) {
        if (line.empty()) continue;
        std::istringstream ss(line.substr(1));
        if (line[0] == 'E' || line[0] == 'L' || line[0] == 'B') {
            std::string hex;
            ss >> hex;
            const std::string text = FromHex(hex);
            std::vector<int> ids;
            if (line[0] == 'E') sp.Encode(text.data(), text.size(), &ids);
            else if (line[0] == 'B') tok_b->Encode(text.data(), (uint32_t)text.size(), &ids);
            else tok->Encode(text.data(), (uint32_t)text.size(), &ids);
            for (size_t i = 0; i < ids.size(); ++i) std::cout << (i ? " " : "") << ids[i];
            std::cout << std::endl;
        } else if (line[0] == 'D' || line[0] == 'T') {
            std::vector<int> ids;
            int v;
            while (ss >> v) ids.push_back(v);
            std::string out;
            if (line[0] == 'D') sp.Decode(ids.data(), ids.size(), &out);
            else tok->Decode(ids.data(), (uint32_t)ids.size(), &out);
            std::cout << "x" << ToHex(out) << std::endl;
        }
    }
    return 0;
}
