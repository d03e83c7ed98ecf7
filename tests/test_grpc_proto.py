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


def test_ctypes_mirrors_of_the_serving_abi(tmp_path):
    """Config / CRequest / CResponse of grpc_server.py are the structs of src/capi/serving_c.h: sizes and the offsets of the fields
    added for the text path (tokenizer inside the C++ generator), taken from a C program compiled against the header"""
    import ctypes
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "capi/serving_c.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(pplsrv_config), sizeof(pplsrv_request), '
                   'sizeof(pplsrv_response), offsetof(pplsrv_config, tokenizer_path), offsetof(pplsrv_config, quant_method), '
                   'offsetof(pplsrv_request, prompt), offsetof(pplsrv_response, text_off)); return 0; }\n')
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "ppl.llm.serving_amd", "src"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [ctypes.sizeof(gs.Config), ctypes.sizeof(gs.CRequest), ctypes.sizeof(gs.CResponse), gs.Config.tokenizer_path.offset,
            gs.Config.quant_method.offset, gs.CRequest.prompt.offset, gs.CResponse.text_off.offset]
    assert got == want
