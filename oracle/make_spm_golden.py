#!/usr/bin/env python3
"""Generates the tokenizer fixtures (runs only in the build container: needs the `sentencepiece` Python module -- the same
library the reference links in C++, src/tokenizer/tokenizer_impl_sp.h).

  tests/golden/spm_bpe.model        LLaMA-style: BPE, byte fallback, identity normalisation, dummy prefix, extra whitespace kept
  tests/golden/spm_unigram.model    unigram, byte fallback, identity normalisation, sentencepiece's default whitespace clean-up
  tests/golden/spm_unigram_nofb.model  unigram WITHOUT byte fallback (unknown characters -> <unk>)
  tests/golden/spm_cases.json       per model: texts with their EncodeAsIds, whole-sequence Decode, per-token Decode, and the
                                    decodes of hand-made id sequences (control pieces, split multi-byte characters, unknowns)
The models are data (a few KB each), trained here on the in-script corpus below; tests/test_tokenizer.py checks the C++
implementation (ppl.llm.serving_amd/src/tokenizer) against the recorded outputs."""
import io
import json
import os

import sentencepiece as spm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")

CORPUS = """The president of the United States is the head of state and head of government.
Hello, my name is Llama and I like to read books about the history of computing.
The capital of France is Paris. The future of AI is being written today, one token at a time.
def fibonacci(n): return n if n < 2 else fibonacci(n - 1) + fibonacci(n - 2)
for i in range(10): print(i, i * i, i ** 3)  # squares and cubes
SELECT name, COUNT(*) FROM users WHERE age > 21 GROUP BY name ORDER BY 2 DESC;
Les mots français avec des accents: été, naïve, cœur, façade, où, déjà vu.
Grüße aus München! Die Straße ist naß und der Käse ist köstlich.
El niño comió piñas en la mañana; ¿dónde está el baño?
数学は美しい。東京は日本の首都です。我喜欢学习新的语言。
Привет, мир! Как дела? Это тест токенизатора.
In 2024 the GPU had 288 GB of HBM3E and 8 TB/s of bandwidth; 256 CUs in 8 XCDs.
tokens per second, time to first token, batch size 1024, sequence length 1024, tensor parallel 8
""" * 4

TEXTS = ["Hello, my name is", "The president of the United States is", " leading space", "trailing space ", "two  spaces   three",
         "", " ", "\n", "tab\tseparated\tvalues", "naïve café déjà vu", "数学は美しい", "emoji 🙂 and 𝔘𝔫𝔦𝔠𝔬𝔡𝔢", "price: $1,234.56 (approx.)",
         "line one\nline two\r\nline three", "ÀÈÌÒÙ àèìòù ÄËÏÖÜ", "x" * 40, "a b c d e f g", "Zażółć gęślą jaźń", "مرحبا بالعالم",
         "mixed 日本語 and English 123", "▁already escaped", "snake_case and camelCase and kebab-case",
         # literal U+2581 in the input: the trailing clean-up strips the space SYMBOL, the merge rule only real spaces
         "ends with the symbol▁", "▁", "▁▁ x ▁ ▁", "a ▁ b", "  ▁  both ▁▁", "δ▁"]


def train(name, **kw):
    model = io.BytesIO()
    spm.SentencePieceTrainer.train(sentence_iterator=iter(CORPUS.splitlines()), model_writer=model, vocab_size=kw.pop("vocab_size", 420),
                                   character_coverage=kw.pop("character_coverage", 0.98), normalization_rule_name="identity",
                                   bos_id=1, eos_id=2, unk_id=0, pad_id=-1, **kw)
    path = os.path.join(OUT, name)
    open(path, "wb").write(model.getvalue())
    return path


def cases(path):
    sp = spm.SentencePieceProcessor(model_file=path)
    out = {"vocab_size": sp.get_piece_size(), "bos": sp.bos_id(), "eos": sp.eos_id(), "unk": sp.unk_id(), "texts": [], "id_sequences": []}
    for t in TEXTS:
        ids = sp.encode(t)
        out["texts"].append({"text": t, "ids": ids, "decoded": sp.decode(ids), "per_token": [sp.decode([i]) for i in ids],
                             "pieces": [sp.id_to_piece(i) for i in ids]})
    byte_ids = [i for i in range(sp.get_piece_size()) if sp.is_byte(i)]
    seqs = [[sp.bos_id()] + sp.encode("Hello world") + [sp.eos_id()], [sp.unk_id()], sp.encode("a") + [sp.unk_id()] + sp.encode("b")]
    if byte_ids:
        b = lambda v: sp.piece_to_id("<0x%02X>" % v)
        e4b896 = [b(0xE4), b(0xB8), b(0x96)]                         # the three bytes of U+4E16
        seqs += [e4b896, e4b896[:1], e4b896[:2], e4b896[1:], [b(0xFF)], sp.encode("x") + e4b896[:2] + sp.encode("y"),
                 [b(0xF0), b(0x9F), b(0x99), b(0x82)], [b(0xF0), b(0x9F)], [b(0xC0), b(0x80)], [b(0xED), b(0xA0), b(0x80)]]
    for s in seqs:
        out["id_sequences"].append({"ids": s, "decoded": sp.decode(s)})
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    res = {}
    p = train("spm_bpe.model", model_type="bpe", byte_fallback=True, add_dummy_prefix=True, remove_extra_whitespaces=False,
              split_digits=True, allow_whitespace_only_pieces=True)
    res["spm_bpe.model"] = cases(p)
    p = train("spm_unigram.model", model_type="unigram", byte_fallback=True)
    res["spm_unigram.model"] = cases(p)
    p = train("spm_unigram_nofb.model", model_type="unigram", byte_fallback=False, vocab_size=240)
    res["spm_unigram_nofb.model"] = cases(p)
    json.dump(res, open(os.path.join(OUT, "spm_cases.json"), "w"), ensure_ascii=True, indent=0)
    print({k: v["vocab_size"] for k, v in res.items()})


if __name__ == "__main__":
    main()
