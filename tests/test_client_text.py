"""CPU checks of the text-mode load generator's host logic (serving/client_qps_measure.py <-> reference tools/client_qps_measure.cc):
request sampling from the conversation-format dataset and the reference's quantile rule."""
import importlib.util
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(os.path.dirname(HERE), "ppl.llm.serving_amd")


def load_client():
    pytest.importorskip("grpc")
    spec = importlib.util.spec_from_file_location("client_qps_measure", os.path.join(PKG, "serving", "client_qps_measure.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_sample_requests_counts_tokens_like_the_reference(tmp_path):
    spm = pytest.importorskip("sentencepiece")
    c = load_client()
    sp = spm.SentencePieceProcessor(model_file=os.path.join(HERE, "golden", "spm_bpe.model"))
    data = [{"conversations": [{"from": "human", "value": "Hello there, how are you?"}, {"from": "gpt", "value": "Fine, thanks."}]},
            {"conversations": [{"from": "human", "value": ""}, {"from": "gpt", "value": "x"}]}]
    path = tmp_path / "d.json"
    json.dump(data, open(path, "w"))
    reqs = c.sample_requests(str(path), sp)
    # reference :71-74: plain Encode of prompt and answer (no BOS), the answer's length becomes max_new_tokens (:86)
    assert reqs[0] == ("Hello there, how are you?", len(sp.encode("Hello there, how are you?")), len(sp.encode("Fine, thanks.")))
    assert reqs[1][1] == 0 and reqs[1][2] == len(sp.encode("x"))


def test_distribution_is_the_reference_index_rule():
    c = load_client()
    v = [float(x) for x in range(200, 0, -1)]            # unsorted on purpose; sorted: 1..200
    d = c.distribution(v)
    # reference :318-340: list[n / 100], list[n / 10], list[n / 4], list[n / 2], list[n * 3 / 4], list[n * 8 / 10], ...
    assert d["min"] == 1.0 and d["max"] == 200.0
    assert d["1%"] == 3.0 and d["10%"] == 21.0 and d["25%"] == 51.0 and d["50%"] == 101.0
    assert d["75%"] == 151.0 and d["80%"] == 161.0 and d["90%"] == 181.0 and d["95%"] == 191.0 and d["99%"] == 199.0
    assert c.distribution([]) == {k: 0.0 for k, _ in c.QUANTILES}
    assert c.distribution([7.0])["50%"] == 7.0
