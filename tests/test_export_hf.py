"""The HF -> model directory exporter (ppl.llm.serving_amd/tools/export_hf_llama.py, counterpart of the reference's ppl.pmx
export step): a tiny random LlamaForCausalLM is saved with save_pretrained, exported (fp16 / W8A16 / W4A16, TP 1 and 2),
loaded back into the CPU oracle from the containers + params.json and compared with the HF model's own logits."""
import importlib.util
import json
import os
import struct

import numpy as np
import pytest

from oracle import ref

torch = pytest.importorskip("torch")
transformers = pytest.importorskip("transformers")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("export_hf_llama", os.path.join(ROOT, "ppl.llm.serving_amd", "tools", "export_hf_llama.py"))
exp = importlib.util.module_from_spec(spec)
spec.loader.exec_module(exp)


def read_container(path):
    out = {}
    with open(path, "rb") as f:
        assert f.read(8) == b"PPLHIPW1"
        (n,) = struct.unpack("<I", f.read(4))
        for _ in range(n):
            (nl,) = struct.unpack("<I", f.read(4))
            name = f.read(nl).decode()
            (nb,) = struct.unpack("<Q", f.read(8))
            f.seek((64 - f.tell() % 64) % 64, 1)
            out[name] = np.frombuffer(f.read(nb), dtype=np.uint8)
    return out


def make_hf_checkpoint(d):
    """saves a tiny random LLaMA (GQA) under d; returns (dir, prompt, HF last-token logits of the prompt)"""
    from transformers import LlamaConfig, LlamaForCausalLM
    torch.manual_seed(3)
    cfg = LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                      vocab_size=384, rms_norm_eps=1e-5, max_position_embeddings=128, tie_word_embeddings=False)
    cfg._attn_implementation = "eager"
    model = LlamaForCausalLM(cfg).float().eval()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 2:
                p.copy_((torch.randn_like(p) * (0.06 if "embed" not in n else 1.0)).half().float())
            else:
                p.copy_((1.0 + 0.1 * torch.randn_like(p)).half().float())
    model.save_pretrained(str(d), safe_serialization=True)
    prompt = [5, 17, 300, 42, 9, 111, 7]
    with torch.no_grad():
        logits = model(torch.tensor([prompt])).logits[0, -1].numpy().astype(np.float32)
    return str(d), prompt, logits


@pytest.fixture(scope="module")
def hf_dir(tmp_path_factory):
    return make_hf_checkpoint(tmp_path_factory.mktemp("hf"))


def oracle_logits(model_dir, tp, prompt):
    p = json.load(open(os.path.join(model_dir, "params.json")))
    desc = ref.make_desc(hidden_dim=p["hidden_dim"], intermediate_dim=p["intermediate_dim"], num_layers=p["num_layers"],
                         num_heads=p["num_heads"], num_kv_heads=p["num_kv_heads"], vocab_size=p["vocab_size"],
                         max_position=p["max_position"], cache_quant_bit=p["cache_quant_bit"], cache_quant_group=p["cache_quant_group"],
                         cache_layout=p["cache_layout"], cache_mode=p["cache_mode"], page_size=p.get("page_size", 0),
                         weight_quant_bit=p["weight_quant_bit"], weight_quant_group=p["weight_quant_group"],
                         norm_eps=p["norm_eps"], rope_theta=p["rope_theta"])
    models = []
    for r in range(tp):
        rm = ref.RefModel(desc, tp_size=tp, tp_rank=r)
        for name, raw in read_container(os.path.join(model_dir, f"model_slice_{r}", "weights.pplhip")).items():
            rm.set_tensor(name, raw)
        rm.kv_alloc(64)
        models.append(rm)
    st = ref.make_step(np.array(prompt), [0, len(prompt)], [0], [0], 0)
    return ref.forward(models, st)[0]


@pytest.mark.parametrize("tp,quant,tol", [(1, "none", 0.02), (2, "none", 0.02), (1, "w8a16", 0.06), (2, "w8a16", 0.06), (2, "w4a16", 0.5)])
def test_export_round_trip(hf_dir, tmp_path, tp, quant, tol):
    d, prompt, want = hf_dir
    out = str(tmp_path / "model")
    exp.main(["--model-dir", d, "--out", out, "--tensor-parallel-size", str(tp), "--quant", quant, "--cache-quant-bit", "0"])
    got = oracle_logits(out, tp, prompt)
    scale = max(1.0, np.abs(want).max())
    assert np.abs(got - want).max() <= tol * scale, (np.abs(got - want).max(), scale)
    if quant != "w4a16":
        srt = np.sort(want)
        if srt[-1] - srt[-2] > 2 * tol * scale:
            assert got.argmax() == want.argmax()


def test_quantisers_known_answers():
    w = np.array([[1.0, -2.0, 0.5, 127.0], [0.0, 0.0, 0.0, 0.0]], dtype=np.float32)
    q, s = exp.quant_w8(w)
    assert q[0].tolist() == [1, -2, 0, 127] or q[0].tolist() == [1, -2, 1, 127]  # 0.5 rounds to even (0)
    assert float(s[0]) == 1.0 and float(s[1]) == 0.0 and (q[1] == 0).all()
    w4 = np.tile(np.array([-8.0, 7.0, 0.0, 3.5], dtype=np.float32), 32)[None, :] * (1.0 / 7.0) * 7.0   # group 128, max |w| = 8
    packed, sc = exp.quant_w4(w4, 128)
    scale = float(sc[0, 0])
    assert abs(scale - 8.0 / 7.0) < 1e-3
    nib = np.stack([packed[0] & 15, packed[0] >> 4], 1).reshape(-1).astype(np.int32) - 8
    deq = nib * scale
    assert np.abs(deq - w4[0]).max() <= scale / 2 + 1e-6
    assert packed.shape == (1, 64) and sc.shape == (1, 1)
