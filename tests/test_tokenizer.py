"""The C++ SentencePiece reader / encoder / decoder (ppl.llm.serving_amd/src/tokenizer; the reference links google/sentencepiece,
src/tokenizer/tokenizer_impl_sp.h:31-74) against outputs recorded from the `sentencepiece` Python module -- the same library --
on three models trained in the build container (oracle/make_spm_golden.py): LLaMA-style BPE with byte fallback, unigram with
byte fallback, unigram without.  CPU only."""
import json
import os
import subprocess

import pytest

from tests.conftest import ROOT

PKG = os.path.join(ROOT, "ppl.llm.serving_amd")
GOLD = os.path.join(ROOT, "tests", "golden")
MODELS = ["spm_bpe.model", "spm_unigram.model", "spm_unigram_nofb.model"]


@pytest.fixture(scope="module")
def tool():
    subprocess.check_call(["make", "-s", "-C", PKG, "build/tokenizer_tool"])
    return os.path.join(PKG, "build", "tokenizer_tool")


@pytest.fixture(scope="module")
def cases():
    return json.load(open(os.path.join(GOLD, "spm_cases.json")))


def run(tool, model, lines):
    out = subprocess.run([tool, os.path.join(GOLD, model)], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=60)
    rows = out.stdout.strip().split("\n")
    assert rows[0].startswith("OK "), rows[0]
    return rows[0].split()[1:], rows[1:]


def hexs(s):
    return s.encode("utf-8").hex()


@pytest.mark.parametrize("model", MODELS)
def test_encode_equals_sentencepiece(tool, cases, model):
    c = cases[model]
    head, rows = run(tool, model, ["E " + hexs(t["text"]) for t in c["texts"]])
    assert [int(x) for x in head] == [c["vocab_size"], c["bos"], c["eos"], c["unk"]]
    for t, row in zip(c["texts"], rows):
        got = [int(x) for x in row.split()] if row else []
        assert got == t["ids"], (t["text"], t["pieces"])


@pytest.mark.parametrize("model", MODELS)
def test_decode_equals_sentencepiece(tool, cases, model):
    c = cases[model]
    seqs = [(t["ids"], t["decoded"]) for t in c["texts"]] + [(s["ids"], s["decoded"]) for s in c["id_sequences"]]
    for t in c["texts"]:                                  # every token on its own (what a streaming server decodes)
        seqs += [([i], d) for i, d in zip(t["ids"], t["per_token"])]
    _, rows = run(tool, model, ["D " + " ".join(str(i) for i in ids) for ids, _ in seqs])
    for (ids, want), row in zip(seqs, rows):
        assert bytes.fromhex(row[1:]).decode("utf-8") == want, (ids, want)


def test_llama_policy_bos_and_streaming_pieces(tool, cases):
    """LlamaTokenizer::Encode puts BOS in front (models/llama/llama_tokenizer.h:35-38); the one-token Decode re-inserts the
    space a word-initial piece loses to the dummy-prefix rule (tokenizer_impl_sp.h:53-59: only when the piece starts with U+2581
    and decodes to something that does not start with a space), so streamed pieces concatenate back to the text."""
    c = cases["spm_bpe.model"]
    t = next(x for x in c["texts"] if x["text"] == "The president of the United States is")
    _, rows = run(tool, "spm_bpe.model", ["L " + hexs(t["text"])] + [f"T {i}" for i in t["ids"]])
    assert [int(x) for x in rows[0].split()] == [c["bos"]] + t["ids"]
    streamed = [bytes.fromhex(r[1:]).decode("utf-8") for r in rows[1:]]
    want = [(" " + d) if (p.startswith("\u2581") and d and d[0] != " ") else d for p, d in zip(t["pieces"], t["per_token"])]
    assert streamed == want
    assert "".join(streamed).replace(" ", "") == t["text"].replace(" ", "") and "".join(streamed).startswith(" The")


def test_baichuan_policy_has_no_bos(tool, cases):
    """models/baichuan/baichuan_tokenizer.h encodes the prompt as it is; llama / internlm / llama3 put BOS first"""
    c = cases["spm_bpe.model"]
    t = next(x for x in c["texts"] if x["text"] == "Hello, my name is")
    _, rows = run(tool, "spm_bpe.model", ["B " + hexs(t["text"]), "L " + hexs(t["text"])])
    assert [int(x) for x in rows[0].split()] == t["ids"]
    assert [int(x) for x in rows[1].split()] == [c["bos"]] + t["ids"]


def test_unsupported_models_are_refused(tool, tmp_path):
    bad = tmp_path / "garbage.model"
    bad.write_bytes(b"\x00\x01\x02not a model")
    out = subprocess.run([tool, str(bad)], input="", capture_output=True, text=True)
    assert out.stdout.startswith("ERROR")


@pytest.mark.parametrize("model", ["spm_bpe.model"])
def test_literal_special_token_text_never_becomes_a_reserved_id(tool, cases, model):
    """BPE merges must run through NORMAL / USER_DEFINED pieces only, so the literal TEXT "<s>", "</s>", "<unk>" or
    "<0x0A>" is tokenised character by character (bos / eos / unk / byte ids cannot be injected from text); compared with the
    sentencepiece module when it is importable."""
    texts = ["<s>", "</s> trailing", "a<unk>b", "<0x0A>", "x <s> y </s>"]
    _, rows = run(tool, model, ["E " + hexs(t) for t in texts])
    c = cases[model]
    got = [[int(x) for x in r.split()] for r in rows]
    for ids in got:
        assert c["bos"] not in ids and c["eos"] not in ids
    try:
        import sentencepiece as spm
    except Exception:
        return
    sp = spm.SentencePieceProcessor(model_file=os.path.join(GOLD, model))
    for t, ids in zip(texts, got):
        assert ids == sp.encode(t), t
