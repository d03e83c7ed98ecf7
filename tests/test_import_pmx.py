"""ppl.pmx export -> model directory importer (ppl.llm.serving_amd/tools/import_pmx_onnx.py).

No ppl.pmx export exists in this environment (parity unpinned), so the test writes the `model_slice_<rank>/model.onnx`
files itself -- protobuf wire format by hand, initializers under ppl.pmx's parameter names, q/k rows in Meta's interleaved
RoPE pairing, tensors inline (`raw_data`, packed `float_data`, fp16 bit patterns in `int32_data`) and as external data --
and checks that the imported containers equal, byte for byte, what tools/export_hf_llama.py produces from the same
weights in HF naming."""
import importlib.util
import json
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = os.path.join(ROOT, "ppl.llm.serving_amd", "tools")
sys.path.insert(0, TOOLS)


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(TOOLS, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


imp = _load("import_pmx_onnx")
exp = _load("export_hf_llama")


# ---- a minimal protobuf writer (test side only) ----------------------------------------------------------------
def varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def fld(num, wt, payload):
    return varint((num << 3) | wt) + (varint(len(payload)) + payload if wt == 2 else payload)


def tensor_proto(name, arr, form, ext_file=None, ext_off=0):
    """TensorProto (onnx.proto:479-602) in one of the encodings torch / onnx tooling produce"""
    dt = {np.dtype(np.float32): 1, np.dtype(np.float16): 10, np.dtype(np.int64): 7}[arr.dtype]
    if form.startswith("as_dtype:"):      # raw bytes stored under another ONNX data type (16 = bfloat16, 17 = float8 ...)
        dt, form = int(form.split(":")[1]), "raw"
    dims = b"".join(fld(1, 0, varint(d)) for d in arr.shape) if form != "packed_dims" else fld(1, 2, b"".join(varint(d) for d in arr.shape))
    body = dims + fld(2, 0, varint(dt)) + fld(8, 2, name.encode())
    if form in ("raw", "packed_dims"):
        body += fld(9, 2, arr.tobytes())
    elif form == "float_data":
        assert arr.dtype == np.float32
        body += fld(4, 2, arr.tobytes())
    elif form == "int32_fp16":
        assert arr.dtype == np.float16
        body += fld(5, 2, b"".join(varint(int(v)) for v in arr.view(np.uint16).ravel()))
    elif form == "external":
        kv = lambda k, v: fld(13, 2, fld(1, 2, k.encode()) + fld(2, 2, str(v).encode()))
        body += kv("location", ext_file) + kv("offset", ext_off) + kv("length", arr.nbytes) + fld(14, 0, varint(1))
    return body


def write_onnx(path, tensors, forms):
    """ModelProto{ir_version, producer_name, graph{node, name, initializer...}} + one external-data file"""
    ext_name, ext = "weights.bin", bytearray()
    inits = b""
    for (name, arr), form in zip(tensors.items(), forms):
        if form == "external":
            ext += b"\0" * ((-len(ext)) % 64)
            inits += fld(5, 2, tensor_proto(name, arr, form, ext_name, len(ext)))
            ext += arr.tobytes()
        else:
            inits += fld(5, 2, tensor_proto(name, arr, form))
    node = fld(1, 2, b"x") + fld(2, 2, b"y") + fld(3, 2, b"n0") + fld(4, 2, b"RMSNorm") + fld(7, 2, b"opmx")
    graph = fld(1, 2, node) + fld(2, 2, b"llama") + inits
    model = fld(1, 0, varint(8)) + fld(2, 2, b"pytorch") + fld(7, 2, graph) + fld(8, 2, fld(1, 2, b"opmx") + fld(2, 0, varint(1)))
    open(path, "wb").write(model)
    open(os.path.join(os.path.dirname(path), ext_name), "wb").write(bytes(ext))


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
            out[name] = f.read(nb)
    return out


def half_to_interleaved(w, n_heads, D):
    """inverse of the importer's row permutation: HF rows (h, t, i) -> Meta rows (h, i, t)"""
    w = w.reshape(n_heads, 2, D // 2, -1)
    return np.stack([w[:, 0], w[:, 1]], axis=2).reshape(n_heads * D, -1)


CFG = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, vocab_size=320)


def hf_state_dict(seed=0):
    rng = np.random.RandomState(seed)
    c = CFG
    D = c["hidden_size"] // c["num_attention_heads"]
    r16 = lambda *s: (rng.randn(*s) * 0.05).astype(np.float16).astype(np.float32)
    sd = {"model.embed_tokens.weight": r16(c["vocab_size"], c["hidden_size"]), "model.norm.weight": 1 + r16(c["hidden_size"]),
          "lm_head.weight": r16(c["vocab_size"], c["hidden_size"])}
    for l in range(c["num_hidden_layers"]):
        p = f"model.layers.{l}."
        sd[p + "input_layernorm.weight"] = 1 + r16(c["hidden_size"])
        sd[p + "post_attention_layernorm.weight"] = 1 + r16(c["hidden_size"])
        sd[p + "self_attn.q_proj.weight"] = r16(c["num_attention_heads"] * D, c["hidden_size"])
        sd[p + "self_attn.k_proj.weight"] = r16(c["num_key_value_heads"] * D, c["hidden_size"])
        sd[p + "self_attn.v_proj.weight"] = r16(c["num_key_value_heads"] * D, c["hidden_size"])
        sd[p + "self_attn.o_proj.weight"] = r16(c["hidden_size"], c["num_attention_heads"] * D)
        sd[p + "mlp.gate_proj.weight"] = r16(c["intermediate_size"], c["hidden_size"])
        sd[p + "mlp.up_proj.weight"] = r16(c["intermediate_size"], c["hidden_size"])
        sd[p + "mlp.down_proj.weight"] = r16(c["hidden_size"], c["intermediate_size"])
    return sd


def pmx_slices(sd, tp, fused_qkv, parallel_embedding):
    """the same weights as ppl.pmx lays them out: per-rank slices, Meta q/k row order, pmx parameter names, fp16"""
    c = CFG
    H, Hkv, D = c["num_attention_heads"], c["num_key_value_heads"], c["hidden_size"] // c["num_attention_heads"]
    h, hk, it, vl, hl = H // tp, Hkv // tp, c["intermediate_size"] // tp, c["vocab_size"] // tp, c["hidden_size"] // tp
    f16 = lambda a: np.ascontiguousarray(a).astype(np.float16)
    out = []
    for r in range(tp):
        emb = sd["model.embed_tokens.weight"]
        t = {"tok_embeddings.weight": f16(emb[:, r * hl:(r + 1) * hl] if parallel_embedding else emb),
             "norm.weight": f16(sd["model.norm.weight"]), "output.weight": f16(sd["lm_head.weight"][r * vl:(r + 1) * vl])}
        for l in range(c["num_hidden_layers"]):
            p, q = f"model.layers.{l}.", f"layers.{l}."
            t[q + "attention_norm.weight"] = f16(sd[p + "input_layernorm.weight"])
            t[q + "ffn_norm.weight"] = f16(sd[p + "post_attention_layernorm.weight"])
            wq = half_to_interleaved(sd[p + "self_attn.q_proj.weight"][r * h * D:(r + 1) * h * D], h, D)
            wk = half_to_interleaved(sd[p + "self_attn.k_proj.weight"][r * hk * D:(r + 1) * hk * D], hk, D)
            wv = sd[p + "self_attn.v_proj.weight"][r * hk * D:(r + 1) * hk * D]
            if fused_qkv:
                t[q + "attention.wqkv.weight"] = f16(np.concatenate([wq, wk, wv], 0))
            else:
                t[q + "attention.wq.weight"], t[q + "attention.wk.weight"], t[q + "attention.wv.weight"] = f16(wq), f16(wk), f16(wv)
            t[q + "attention.wo.weight"] = f16(sd[p + "self_attn.o_proj.weight"][:, r * h * D:(r + 1) * h * D])
            t[q + "feed_forward.w1.weight"] = f16(sd[p + "mlp.gate_proj.weight"][r * it:(r + 1) * it])
            t[q + "feed_forward.w3.weight"] = f16(sd[p + "mlp.up_proj.weight"][r * it:(r + 1) * it])
            t[q + "feed_forward.w2.weight"] = f16(sd[p + "mlp.down_proj.weight"][:, r * it:(r + 1) * it])
        out.append(t)
    return out


def test_rope_row_permutation_is_the_inverse_of_metas_interleave():
    w = np.arange(2 * 8 * 3, dtype=np.float32).reshape(16, 3)
    m = half_to_interleaved(w, 2, 8)
    assert (imp.rope_rows_interleaved_to_half(m, 2, 8) == w).all()
    # head 0: HF row i pairs with row i + 4; in Meta order they sit next to each other
    assert (m[0] == w[0]).all() and (m[1] == w[4]).all() and (m[2] == w[1]).all()


@pytest.mark.parametrize("tp,fused_qkv,parallel_embedding,quant", [(1, True, False, "w8a16"), (2, True, True, "none"),
                                                                    (2, False, False, "w4a16")])
def test_import_equals_hf_export_of_the_same_weights(tmp_path, tp, fused_qkv, parallel_embedding, quant):
    sd = hf_state_dict(tp)
    src = tmp_path / "pmx"
    forms_cycle = ["raw", "external", "packed_dims", "int32_fp16", "external"]
    for r, t in enumerate(pmx_slices(sd, tp, fused_qkv, parallel_embedding)):
        d = src / f"model_slice_{r}"
        os.makedirs(d)
        tensors = dict(t)
        tensors["onnx::unused_const"] = np.array([1, 2, 3], dtype=np.int64)      # things a real graph also carries
        tensors["rope.inv_freq"] = np.linspace(0, 1, 8, dtype=np.float32)
        forms = [forms_cycle[i % len(forms_cycle)] if a.dtype == np.float16 else ("float_data" if a.dtype == np.float32 else "raw")
                 for i, a in enumerate(tensors.values())]
        write_onnx(str(d / "model.onnx"), tensors, forms)
    params = {"num_heads": 4, "num_kv_heads": 2, "num_layers": 2, "hidden_dim": 256, "intermediate_dim": 512, "vocab_size": 320,
              "cache_quant_bit": 8, "cache_quant_group": 8, "cache_layout": 3, "cache_mode": 0, "dynamic_batching": True, "auto_causal": True}
    json.dump(params, open(src / "params.json", "w"))
    out = tmp_path / "hip"
    assert imp.main(["--model-dir", str(src), "--out", str(out), "--quant", quant, "--quant-group", "128"]) is None
    want = exp.convert(sd, CFG, tp, quant, 128)
    for r in range(tp):
        got = read_container(out / f"model_slice_{r}" / "weights.pplhip")
        assert sorted(got) == sorted(want[r])
        for name, arr in want[r].items():
            assert got[name] == np.ascontiguousarray(arr).tobytes(), (r, name)
    p = json.load(open(out / "params.json"))
    assert p["weight_quant_bit"] == {"none": 0, "w8a16": 8, "w4a16": 4}[quant] and p["num_kv_heads"] == 2 and p["cache_layout"] == 3


def test_errors_name_the_missing_piece(tmp_path):
    sd = hf_state_dict(5)
    t = pmx_slices(sd, 1, True, False)[0]
    del t["layers.1.attention.wo.weight"]
    d = tmp_path / "pmx" / "model_slice_0"
    os.makedirs(d)
    write_onnx(str(d / "model.onnx"), t, ["raw"] * len(t))
    json.dump({"num_heads": 4, "num_kv_heads": 2, "num_layers": 2, "hidden_dim": 256, "intermediate_dim": 512, "vocab_size": 320},
              open(tmp_path / "pmx" / "params.json", "w"))
    with pytest.raises(KeyError, match="layers.1.attention.wo.weight"):
        imp.convert_dir(str(tmp_path / "pmx"), str(tmp_path / "out"))
    os.rename(tmp_path / "pmx" / "model_slice_0", tmp_path / "pmx" / "model_slice_1")
    with pytest.raises(ValueError, match="model_slice_0"):
        imp.convert_dir(str(tmp_path / "pmx"), str(tmp_path / "out"))


def test_bfloat16_initializers_are_read_and_unsupported_ones_are_named(tmp_path):
    """a bf16 export loads (bit patterns widened to float32, exact); an initializer in a type the importer cannot read is
    reported with its name and ONNX type when a weight needs it -- not as a bare "no initializer"."""
    sd = hf_state_dict(6)
    t = pmx_slices(sd, 1, True, False)[0]
    d = tmp_path / "pmx" / "model_slice_0"
    os.makedirs(d)
    forms = []
    bf = {}
    for name, arr in t.items():
        if name.endswith("attention.wo.weight"):                        # store as bfloat16 bit patterns
            hi = (arr.astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
            bf[name] = (hi.astype(np.uint32) << 16).view(np.float32)
            t[name] = hi.view(np.float16)                               # (tensor_proto keys on the numpy dtype; bytes are what matter)
            forms.append("as_dtype:16")
        else:
            forms.append("raw")
    write_onnx(str(d / "model.onnx"), t, forms)
    got = imp.read_initializers(str(d / "model.onnx"))
    for name, want in bf.items():
        assert got[name].dtype == np.float32 and (got[name] == want).all()
    forms[list(t).index("layers.0.attention.wo.weight")] = "as_dtype:17"   # float8: unreadable
    write_onnx(str(d / "model.onnx"), t, forms)
    json.dump({"num_heads": 4, "num_kv_heads": 2, "num_layers": 2, "hidden_dim": 256, "intermediate_dim": 512, "vocab_size": 320},
              open(tmp_path / "pmx" / "params.json", "w"))
    with pytest.raises(TypeError, match="layers.0.attention.wo.weight.*FLOAT8E4M3FN"):
        imp.convert_dir(str(tmp_path / "pmx"), str(tmp_path / "out"))
