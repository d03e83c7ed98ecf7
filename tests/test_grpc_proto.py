"""Wire format and request parsing of the gRPC front end (no device, no server): the runtime descriptors of llm_proto.py
must produce the bytes protoc-generated code would for src/serving/grpc/proto/llm.proto, and parse_request must apply the
defaults of grpc_server.cc:218-252."""
import os
import sys

import pytest

pytest.importorskip("grpc")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ppl.llm.serving_amd", "serving"))
import grpc_server as gs  # noqa: E402
import llm_proto as P  # noqa: E402


def test_wire_bytes_known_answers():
    r = P.BatchedRequest()
    q = r.req.add()
    q.id = 7
    q.tokens.ids.extend([1, 2, 300])
    q.stopping_parameters.max_new_tokens = 5
    q.stopping_parameters.ignore_eos_token = True
    # hand-encoded proto3: req(1,len) { id(1)=7  tokens(3,len){ ids(1,packed)=01 02 ac02 }  stopping(5,len){ max_new(1)=5 ignore(3)=1 } }
    assert r.SerializeToString().hex() == "0a1008071a060a040102ac022a0408051801"
    rsp = P.BatchedResponse()
    x = rsp.rsp.add()
    x.status, x.id = P.FINISHED, 9
    x.tokens.ids.append(42)
    x.detail.logprobs = -0.5
    x.detail.finish_reason = 1
    # rsp(1,len){ status(1)=1 id(2)=9 tokens(4,len){0a 01 2a} detail(5,len){ logprobs(1,fixed32)=-0.5 finish_reason(3)=1 } }
    assert rsp.SerializeToString().hex() == "0a120801100922030a012a2a070d000000bf1801"
    back = P.BatchedResponse.FromString(rsp.SerializeToString())
    assert back.rsp[0].id == 9 and list(back.rsp[0].tokens.ids) == [42] and back.rsp[0].detail.logprobs == -0.5
    assert P.METHOD == "/ppl.llm.proto.LLMService/Generation"


def test_parse_request_defaults():
    pb = P.Request()
    pb.stopping_parameters.max_new_tokens = 12
    kw = gs.parse_request(pb)                       # nothing set: greedy, temperature 1, repetition penalty 1, EOS honoured
    assert kw == dict(temperature=1.0, top_k=1, top_p=0.0, repetition_penalty=1.0, presence_penalty=0.0, frequency_penalty=0.0,
                      generation_length=12, early_stopping=1)
    pb.choosing_parameters.do_sample = True
    pb.choosing_parameters.top_k = 40
    pb.choosing_parameters.top_p = 1.5              # outside [0, 1] -> 0
    pb.choosing_parameters.temperature = 0.5
    pb.choosing_parameters.repetition_penalty = 1.25
    pb.stopping_parameters.ignore_eos_token = True
    kw = gs.parse_request(pb)
    assert kw["top_k"] == 40 and kw["top_p"] == 0.0 and kw["temperature"] == 0.5 and kw["repetition_penalty"] == 1.25
    assert kw["early_stopping"] == 0
    pb.choosing_parameters.top_p = 0.75
    pb.choosing_parameters.do_sample = False        # sampling off overrides top_k / top_p
    kw = gs.parse_request(pb)
    assert kw["top_k"] == 1 and kw["top_p"] == 0.0


def test_tokenizer_text_path(tmp_path):
    """sentencepiece text path of the server: BOS first, and a decoded piece gets its leading space back when the piece
    starts with U+2581 (src/tokenizer/tokenizer_impl_sp.h:53-59)"""
    spm = pytest.importorskip("sentencepiece")
    corpus = tmp_path / "corpus.txt"
    corpus.write_text("\n".join(["the quick brown fox jumps over the lazy dog", "hello world this is a tokenizer test",
                                 "the president of the united states", "the capital of france is paris"] * 20))
    spm.SentencePieceTrainer.train(input=str(corpus), model_prefix=str(tmp_path / "tok"), vocab_size=64, model_type="bpe",
                                   bos_id=1, eos_id=2, unk_id=0, minloglevel=2)
    tok = gs.Tokenizer(str(tmp_path / "tok.model"))
    ids = tok.encode("the quick fox")
    assert ids[0] == 1 and len(ids) > 1
    text = "".join(tok.decode_one(t) for t in ids[1:])
    assert text.strip() == "the quick fox"
    assert text.startswith(" ")          # the first word piece carries the U+2581 marker
