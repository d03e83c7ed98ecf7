"""The gRPC front end on the GPU: serving/grpc_server.py (wire format of the reference's llm.proto) in a subprocess, a raw
grpc client in the test.  Greedy token streams must equal the oracle's continuation; a rejected request answers FAILED; the
load generator (client_qps_measure_token_in_out.py) runs against the same server."""
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import pytest

from oracle import ref
from tests.conftest import ROOT
from tests.test_gpu_tools import CFG, PKG, oracle_greedy

pytestmark = pytest.mark.gpu
grpc = pytest.importorskip("grpc")
sys.path.insert(0, os.path.join(PKG, "serving"))
import llm_proto as P  # noqa: E402


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def server():
    assert os.path.exists(os.path.join(PKG, "build", "libpplserving_c.so")), "run __graft_entry__.build()"
    port = free_port()
    proc = subprocess.Popen([sys.executable, os.path.join(PKG, "serving", "grpc_server.py"), "--model-param-path", CFG,
                             "--synthetic-weights", "--synthetic-seed", "77", "--kv-cache-max-tokens", "2048", "--max-running-batch", "16",
                             "--max-tokens-per-step", "256", "--max-input-tokens-per-request", "64", "--host", "127.0.0.1",
                             "--port", str(port)], stderr=subprocess.PIPE, text=True)
    t0 = time.time()
    line = ""
    while time.time() - t0 < 120:
        line = proc.stderr.readline()
        if "listening" in line or proc.poll() is not None:
            break
    assert "listening" in line, f"server did not start: {line}"
    yield f"127.0.0.1:{port}"
    proc.terminate()
    try:
        proc.wait(timeout=20)
    except subprocess.TimeoutExpired:
        proc.kill()


def call(target, reqs):
    """reqs: list of (id, tokens, max_new_tokens); returns {id: (tokens, statuses, finish_reason)}"""
    out = {}
    with grpc.insecure_channel(target) as ch:
        stub = ch.unary_stream(P.METHOD, request_serializer=P.BatchedRequest.SerializeToString,
                               response_deserializer=P.BatchedResponse.FromString)
        br = P.BatchedRequest()
        for rid, toks, n in reqs:
            r = br.req.add()
            r.id = rid
            r.tokens.ids.extend(toks)
            r.stopping_parameters.max_new_tokens = n
            r.stopping_parameters.ignore_eos_token = True
        for batch in stub(br, timeout=120):
            for rsp in batch.rsp:
                rec = out.setdefault(rsp.id, ([], [], []))
                rec[1].append(rsp.status)
                if rsp.status != P.FAILED:
                    rec[0].extend(rsp.tokens.ids)
                    rec[2].append(rsp.detail.finish_reason)
    return out


def test_streams_equal_the_oracle(server):
    cfg = json.load(open(CFG))
    desc = ref.make_desc(hidden_dim=cfg["hidden_dim"], intermediate_dim=cfg["intermediate_dim"], num_layers=cfg["num_layers"],
                         num_heads=cfg["num_heads"], num_kv_heads=cfg["num_kv_heads"], vocab_size=cfg["vocab_size"],
                         max_position=cfg["max_position"], cache_quant_bit=8, cache_quant_group=8, cache_layout=3, cache_mode=1,
                         page_size=4, weight_quant_bit=8)
    rng = np.random.RandomState(4)
    reqs = [(100 + i, rng.randint(3, cfg["vocab_size"], size=n).tolist(), g) for i, (n, g) in enumerate([(9, 6), (3, 8), (17, 5)])]
    got = call(server, reqs)
    assert sorted(got) == [100, 101, 102]                                   # the client's ids come back
    compared = 0
    for rid, toks, g in reqs:
        tokens, statuses, reasons = got[rid]
        assert len(tokens) == g and statuses[-1] == P.FINISHED and all(s == P.PROCESSING for s in statuses[:-1])
        assert reasons[-1] == 0                                             # FINISH_REASON_LENGTH
        want, margins = oracle_greedy(desc, 77, toks, g)
        for i, (a, b) in enumerate(zip(tokens, want)):
            if margins[i] < 8e-3:
                break
            assert a == b, (rid, i, tokens, want)
            compared += 1
    assert compared >= 8


def test_rejected_request_answers_failed_and_others_proceed(server):
    got = call(server, [(1, list(range(3, 3 + 100)), 4), (2, [5, 6, 7], 3)])    # 100 tokens > --max-input-tokens-per-request 64
    assert got[1][1] == [P.FAILED] and got[1][0] == []
    assert len(got[2][0]) == 3 and got[2][1][-1] == P.FINISHED


def test_load_generator_against_the_server(server):
    out = subprocess.check_output([sys.executable, os.path.join(PKG, "serving", "client_qps_measure_token_in_out.py"), "--target", server,
                                   "--num-requests", "24", "--request-rate", "200", "--vocab-size", "1024", "--max-seq-len", "64"],
                                  timeout=300).decode()
    res = json.loads(out.strip().splitlines()[-1])
    assert res["failed"] == 0 and res["requests"] == 24 and res["out_tps"] > 0 and res["ttft_ms"]["p50"] > 0


# ---- text requests: tokenised and detokenised inside the C++ generator (src/tokenizer), like the reference's server -------------
@pytest.fixture(scope="module")
def text_server(golden_dir, tmp_path_factory):
    port = free_port()
    cfg = json.load(open(CFG))
    cfg["vocab_size"] = 420                                   # = the fixture tokenizer's vocabulary: every generated id is a piece
    params = str(tmp_path_factory.mktemp("textsrv") / "params.json")
    json.dump(cfg, open(params, "w"))
    proc = subprocess.Popen([sys.executable, os.path.join(PKG, "serving", "grpc_server.py"), "--model-param-path", params,
                             "--synthetic-weights", "--synthetic-seed", "77", "--kv-cache-max-tokens", "2048", "--max-running-batch", "16",
                             "--max-tokens-per-step", "256", "--host", "127.0.0.1", "--port", str(port),
                             "--tokenizer-path", os.path.join(golden_dir, "spm_bpe.model")], stderr=subprocess.PIPE, text=True)
    t0 = time.time()
    line = ""
    while time.time() - t0 < 120:
        line = proc.stderr.readline()
        if "listening" in line or proc.poll() is not None:
            break
    assert "listening" in line, f"server did not start: {line}"
    yield f"127.0.0.1:{port}"
    proc.terminate()
    try:
        proc.wait(timeout=20)
    except subprocess.TimeoutExpired:
        proc.kill()


def reference_stream(sp, toks):
    """DecodeAndSendTask (src/generator/llm_generator.cc:58-112) + SentencePieceTokenizer::Decode (tokenizer_impl_sp.h:53-59),
    restated with the sentencepiece module: one response text per generated token"""
    out, flag, buf = [], 0, [0, 0, 0]
    for t in toks:
        s = sp.decode([int(t)])
        if sp.id_to_piece(int(t)).startswith("▁") and s and s[0] != " ":
            s = " " + s
        if s == "�" and flag < 3:
            buf[flag] = int(t)
            flag += 1
            s = ""
            if flag == 3:
                s = sp.decode(buf)
                flag, buf = 0, [0, 0, 0]
        out.append(s)
    return out


def test_text_requests_stream_what_the_reference_rule_gives(text_server, golden_dir):
    spm = pytest.importorskip("sentencepiece")
    sp = spm.SentencePieceProcessor(model_file=os.path.join(golden_dir, "spm_bpe.model"))
    assert sp.get_piece_size() == 420
    prompts = ["Hello, my name is", "The president of the United States is", "naïve café 数学 🙂"]
    n_new = 24
    with grpc.insecure_channel(text_server) as ch:
        stub = ch.unary_stream(P.METHOD, request_serializer=P.BatchedRequest.SerializeToString,
                               response_deserializer=P.BatchedResponse.FromString)
        br = P.BatchedRequest()
        for i, text in enumerate(prompts):
            r = br.req.add()                                  # text request
            r.id, r.prompt = i, text
            r.stopping_parameters.max_new_tokens = n_new
            r.stopping_parameters.ignore_eos_token = True
            r = br.req.add()                                  # the same prompt as tokens: LlamaTokenizer = BOS + pieces
            r.id = 100 + i
            r.tokens.ids.extend([sp.bos_id()] + sp.encode(text))
            r.stopping_parameters.max_new_tokens = n_new
            r.stopping_parameters.ignore_eos_token = True
        pieces, tokens = {}, {}
        for batch in stub(br, timeout=120):
            for rsp in batch.rsp:
                assert rsp.status != P.FAILED
                if rsp.id >= 100:
                    tokens.setdefault(rsp.id - 100, []).extend(rsp.tokens.ids)
                else:
                    pieces.setdefault(rsp.id, []).append(rsp.generated)
    for i in range(len(prompts)):
        assert len(tokens[i]) == n_new and len(pieces[i]) == n_new
        assert pieces[i] == reference_stream(sp, tokens[i]), (i, pieces[i], tokens[i])


def test_text_mode_load_generator(text_server, golden_dir, tmp_path):
    """serving/client_qps_measure.py = the reference's tools/client_qps_measure.cc: conversation-format dataset, prompts sent as text,
    max_new_tokens = the recorded answer's token count, ignore_eos_token -> every request generates exactly that many tokens."""
    spm = pytest.importorskip("sentencepiece")
    sp = spm.SentencePieceProcessor(model_file=os.path.join(golden_dir, "spm_bpe.model"))
    convs = [("Hello, my name is", "I am a small model and this is what I say."), ("The president of the United States is", "someone"),
             ("naïve café 数学 🙂", "yes " * 9), ("What is 2 + 2?", "It is four, as far as anyone can tell."), ("a", "b c d e f g h")]
    data = [{"id": f"c{i}", "conversations": [{"from": "human", "value": p}, {"from": "gpt", "value": a}]} for i, (p, a) in enumerate(convs)]
    ds, dump = str(tmp_path / "samples.json"), str(tmp_path / "answers.json")
    json.dump(data, open(ds, "w"))
    proc = subprocess.run([sys.executable, os.path.join(PKG, "serving", "client_qps_measure.py"), "--target", text_server, "--tokenizer",
                           os.path.join(golden_dir, "spm_bpe.model"), "--dataset", ds, "--request_rate", "50", "--dump-answers", dump],
                          timeout=300, capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr[-2000:]
    res = json.loads(proc.stdout.strip().splitlines()[-1])
    want_in = sum(len(sp.encode(p)) for p, _ in convs)
    want_out = sum(len(sp.encode(a)) for _, a in convs)
    assert res["failed"] == 0 and res["request_count"] == 5
    assert res["total_input_len"] == want_in and res["expected_total_gen_len"] == want_out
    assert res["real_total_gen_len"] == want_out                       # ignore_eos_token: exactly max_new_tokens responses each
    assert res["tokens_out_per_sec"] > 0 and res["prefill_latency_ms"]["50%"] > 0 and res["avg_latency_decoding_ms"] > 0
    assert "[RESULT] tokens out per sec:" in proc.stderr and "[RESULT] prefill latency distribution (ms):" in proc.stderr
    answers = json.load(open(dump))
    assert sorted(answers) == ["0", "1", "2", "3", "4"] and all(isinstance(t, str) for t in answers.values())
