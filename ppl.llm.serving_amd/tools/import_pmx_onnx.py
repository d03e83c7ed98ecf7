#!/usr/bin/env python3
"""Converts a ppl.pmx LLaMA export -- the model directory the reference serves (docs/llama_guide.md:12-38:
`params.json` + `model_slice_<rank>/model.onnx`, loaded by src/backends/cuda/resource_manager.cc:280-289) -- into the
model directory this backend loads (`params.json` + `model_slice_<rank>/weights.pplhip`).
SURVEY.md 8(f) row N2.

Only the weights are taken from model.onnx: its initializers (TensorProto, src/onnx/onnx.proto:479-602 -- inline
`raw_data` or the external-data form torch.onnx.export uses above 2 GB: key/value entries `location`, `offset`, `length`,
onnx.proto:591,602); the graph itself is not interpreted, the forward pass is fixed in this backend.  The file is read with
a small protobuf wire-format reader (no onnx / protobuf package needed, tensors are memory-mapped, not copied).

Initializer names are the torch parameter paths of ppl.pmx's LLaMA model (model_zoo/llama/modeling):
    tok_embeddings.weight, norm.weight, output.weight,
    layers.<l>.attention_norm.weight, layers.<l>.ffn_norm.weight,
    layers.<l>.attention.wqkv.weight            (--fused_qkv 1: this rank's q, k, v rows concatenated)  or  wq / wk / wv,
    layers.<l>.attention.wo.weight,
    layers.<l>.feed_forward.w1 / w3 (gate / up, column parallel), w2 (row parallel)
every slice already cut for its tensor-parallel rank (column-parallel weights on dim 0, row-parallel on dim 1; embedding
and norms whole or, for a ParallelEmbedding sliced on the hidden dim, re-assembled from all ranks).
ppl.pmx exports Meta checkpoints, whose q/k rows are in the interleaved RoPE pairing (2i, 2i+1); this backend implements
the half-split pairing (i, i + D/2) -- DESIGN.md "numerics" -- so q and k rows are permuted per head (`--rope-pairing`).
Quantisation (W8A16 / W4A16) is the same post-export step as tools/export_hf_llama.py.

PARITY UNPINNED: no ppl.pmx export exists in this environment; the reader is tested against ONNX files written by the test
itself with these names (tests/test_import_pmx.py), and the result is checked against export_hf_llama.py on the same weights.

    python import_pmx_onnx.py --model-dir /model_data/llama_7b_ppl --out /model_data/llama_7b_hip --quant w8a16
"""
import argparse
import json
import mmap
import os
import re
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import export_hf_llama as X  # noqa: E402  (container writer and quantisers)

# TensorProto.DataType (onnx.proto:479-507) -> numpy
DTYPES = {1: np.float32, 2: np.uint8, 3: np.int8, 6: np.int32, 7: np.int64, 10: np.float16, 11: np.float64,
          16: np.uint16}   # 16 = BFLOAT16: read as raw 16-bit patterns, widened to float32 below
ONNX_DTYPE_NAMES = {0: "UNDEFINED", 8: "STRING", 9: "BOOL", 12: "UINT32", 13: "UINT64", 14: "COMPLEX64", 15: "COMPLEX128", 17: "FLOAT8E4M3FN",
                    18: "FLOAT8E4M3FNUZ", 19: "FLOAT8E5M2", 20: "FLOAT8E5M2FNUZ", 21: "UINT4", 22: "INT4"}
UNSUPPORTED = {}           # initializer name -> ONNX data type it was stored in (skipped; named if somebody needs it later)


# ---------------------------------------------------------------------------------------------- protobuf wire format
def _varint(buf, pos):
    r = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        r |= (b & 0x7F) << shift
        if not b & 0x80:
            return r, pos
        shift += 7
        if shift > 70:
            raise ValueError("malformed varint")


def fields(buf, pos, end):
    """yields (field_number, wire_type, value): value = int for varint / fixed, (start, end) for length-delimited"""
    while pos < end:
        key, pos = _varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = int.from_bytes(buf[pos:pos + 8], "little")
            pos += 8
        elif wt == 5:
            v = int.from_bytes(buf[pos:pos + 4], "little")
            pos += 4
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v = (pos, pos + n)
            pos += n
        else:
            raise ValueError(f"unsupported wire type {wt} (field {fn})")
        if pos > end:
            raise ValueError("field runs past the end of its message")
        yield fn, wt, v


def _packed_int64(buf, span):
    pos, end = span
    out = []
    while pos < end:
        v, pos = _varint(buf, pos)
        out.append(v - (1 << 64) if v >> 63 else v)
    return out


def read_initializers(path):
    """model.onnx -> {name: np.ndarray} (views into memory maps of the file / its external-data files)"""
    f = open(path, "rb")
    buf = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    graph = None
    for fn, wt, v in fields(buf, 0, len(buf)):
        if fn == 7 and wt == 2:      # ModelProto.graph (onnx.proto:380)
            graph = v
    if graph is None:
        raise ValueError(f"{path}: no graph")
    ext_maps = {}
    out = {}
    for fn, wt, v in fields(buf, *graph):
        if fn != 5 or wt != 2:       # GraphProto.initializer (onnx.proto:450)
            continue
        dims, dtype, name, raw, ext, location = [], 0, "", None, {}, 0
        typed = {}
        for tf, tw, tv in fields(buf, *v):
            if tf == 1:              # dims: packed or one varint per entry
                dims += _packed_int64(buf, tv) if tw == 2 else [tv]
            elif tf == 2:
                dtype = tv
            elif tf == 8:
                name = bytes(buf[tv[0]:tv[1]]).decode()
            elif tf == 9:
                raw = tv
            elif tf == 13:           # StringStringEntryProto (onnx.proto:417-419)
                k = val = ""
                for ef, _, ev in fields(buf, *tv):
                    if ef == 1:
                        k = bytes(buf[ev[0]:ev[1]]).decode()
                    elif ef == 2:
                        val = bytes(buf[ev[0]:ev[1]]).decode()
                ext[k] = val
            elif tf == 14:
                location = tv
            elif tf in (4, 5, 7) and tw == 2:   # float_data / int32_data / int64_data, packed
                typed[tf] = tv
        if dtype not in DTYPES:
            UNSUPPORTED[name] = ONNX_DTYPE_NAMES.get(dtype, str(dtype))   # strings, bools, fp8 ...: reported if a weight needs it
            continue
        np_t = np.dtype(DTYPES[dtype])
        count = int(np.prod(dims)) if dims else 1
        if location == 1 or ext:     # EXTERNAL (onnx.proto:602)
            loc = ext.get("location")
            if not loc:
                raise ValueError(f"{name}: external tensor without a location")
            p = os.path.join(os.path.dirname(path), loc)
            if p not in ext_maps:
                ef_ = open(p, "rb")
                ext_maps[p] = mmap.mmap(ef_.fileno(), 0, access=mmap.ACCESS_READ) if os.path.getsize(p) else b""
            off = int(ext.get("offset", 0))
            arr = np.frombuffer(ext_maps[p], dtype=np_t, count=count, offset=off)
        elif raw is not None:
            arr = np.frombuffer(buf, dtype=np_t, count=count, offset=raw[0])
        elif 4 in typed and dtype == 1:
            arr = np.frombuffer(buf, dtype=np.float32, count=count, offset=typed[4][0])
        elif 7 in typed and dtype == 7:
            arr = np.array(_packed_int64(buf, typed[7]), dtype=np.int64)
        elif 5 in typed:             # int32_data also carries fp16 bit patterns / int8 / uint8 values, one per varint
            vals = np.array(_packed_int64(buf, typed[5]), dtype=np.int64)
            arr = vals.astype(np.uint16).view(np.float16) if dtype == 10 else vals.astype(np_t)
        else:
            arr = np.zeros(count, dtype=np_t)
        if dtype == 16:             # bfloat16 bit patterns -> float32 (exact)
            arr = (arr.astype(np.uint32) << 16).view(np.float32)
        out[name] = arr.reshape(dims)
    return out


# ---------------------------------------------------------------------------------------------- mapping
def rope_rows_interleaved_to_half(w, n_heads, D):
    """rows of a q / k projection [n_heads * D, K]: per head, row 2i -> i and row 2i+1 -> i + D/2"""
    w = w.reshape(n_heads, D // 2, 2, -1)
    return np.concatenate([w[:, :, 0], w[:, :, 1]], axis=1).reshape(n_heads * D, -1)


def convert_slice(sd, p, rank, tp, quant, group, interleaved_rope):
    """one rank's initializers (pmx names) -> container tensors (this backend's names)"""
    H, Hkv = p["num_heads"] // tp, p.get("num_kv_heads", p["num_heads"]) // tp
    hd, L = p["hidden_dim"], p["num_layers"]
    D = hd // p["num_heads"]
    f16 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float16)

    def need(name):
        if name not in sd:
            if name in UNSUPPORTED:
                raise TypeError(f"model_slice_{rank}/model.onnx stores '{name}' as ONNX {UNSUPPORTED[name]}, which this importer cannot read")
            raise KeyError(f"model_slice_{rank}/model.onnx has no initializer '{name}'")
        return np.asarray(sd[name], dtype=np.float32)

    out = {"norm.weight": f16(need("norm.weight")), "output.weight": f16(need("output.weight"))}

    def put(name, w):
        if quant == "none":
            out[name + ".weight"] = f16(w)
        elif quant == "w8a16":
            out[name + ".weight"], out[name + ".scale"] = X.quant_w8(w)
        else:
            out[name + ".weight"], out[name + ".scale"] = X.quant_w4(w, group)

    fix = (lambda w, n: rope_rows_interleaved_to_half(w, n, D)) if interleaved_rope else (lambda w, n: w)
    for l in range(L):
        pre = f"layers.{l}."
        out[pre + "attention_norm.weight"] = f16(need(pre + "attention_norm.weight"))
        out[pre + "ffn_norm.weight"] = f16(need(pre + "ffn_norm.weight"))
        if pre + "attention.wqkv.weight" in sd:
            w = need(pre + "attention.wqkv.weight")
            if w.shape[0] != (H + 2 * Hkv) * D:
                raise ValueError(f"{pre}attention.wqkv.weight has {w.shape[0]} rows, expected {(H + 2 * Hkv) * D} for rank {rank} of {tp}")
            q, k, v = w[:H * D], w[H * D:(H + Hkv) * D], w[(H + Hkv) * D:]
        else:
            q, k, v = need(pre + "attention.wq.weight"), need(pre + "attention.wk.weight"), need(pre + "attention.wv.weight")
        put(pre + "attention.wqkv", np.concatenate([fix(q, H), fix(k, Hkv), v], 0))
        put(pre + "attention.wo", need(pre + "attention.wo.weight"))
        put(pre + "feed_forward.w13", np.concatenate([need(pre + "feed_forward.w1.weight"), need(pre + "feed_forward.w3.weight")], 0))
        put(pre + "feed_forward.w2", need(pre + "feed_forward.w2.weight"))
    return out


def convert_dir(model_dir, out_dir, quant="w8a16", group=128, rope_pairing="interleaved", extra=None):
    p = json.load(open(os.path.join(model_dir, "params.json")))
    ranks = sorted(int(m.group(1)) for m in (re.fullmatch(r"model_slice_(\d+)", d) for d in os.listdir(model_dir)) if m)
    tp = len(ranks)
    if ranks != list(range(tp)) or tp == 0:
        raise ValueError(f"{model_dir}: expected model_slice_0 .. model_slice_<n-1>, found {ranks}")
    sds = [read_initializers(os.path.join(model_dir, f"model_slice_{r}", "model.onnx")) for r in range(tp)]
    # token embedding: whole in every slice, or (ParallelEmbedding) cut on the hidden dim -> concatenate the ranks' columns
    emb = [np.asarray(sd["tok_embeddings.weight"], dtype=np.float32) for sd in sds]
    if emb[0].shape[1] * tp == p["hidden_dim"] and tp > 1:
        emb_full = np.concatenate(emb, axis=1)
    elif emb[0].shape[0] * tp == p["vocab_size"] and tp > 1:
        emb_full = np.concatenate(emb, axis=0)
    else:
        emb_full = emb[0]
    if emb_full.shape != (p["vocab_size"], p["hidden_dim"]):
        raise ValueError(f"tok_embeddings.weight assembles to {emb_full.shape}, expected {(p['vocab_size'], p['hidden_dim'])}")
    os.makedirs(out_dir, exist_ok=True)
    for r in range(tp):
        t = {"tok_embeddings.weight": emb_full.astype(np.float16)}
        t.update(convert_slice(sds[r], p, r, tp, quant, group, rope_pairing == "interleaved"))
        d = os.path.join(out_dir, f"model_slice_{r}")
        os.makedirs(d, exist_ok=True)
        X.write_container(os.path.join(d, "weights.pplhip"), t)
    q = dict(p)   # the reference's keys stay; this build's optional keys are added (src/common/config.h ModelConfig)
    q.update({"weight_quant_bit": {"none": 0, "w8a16": 8, "w4a16": 4}[quant], "weight_quant_group": group})
    q.update(extra or {})
    json.dump(q, open(os.path.join(out_dir, "params.json"), "w"), indent=1)
    return tp


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--model-dir", required=True, help="ppl.pmx export: params.json + model_slice_<rank>/model.onnx")
    ap.add_argument("--out", required=True)
    ap.add_argument("--quant", choices=["none", "w8a16", "w4a16"], default="w8a16")
    ap.add_argument("--quant-group", type=int, default=128)
    ap.add_argument("--rope-pairing", choices=["interleaved", "half"], default="interleaved",
                    help="pairing of the q/k rows in the export (Meta checkpoints: interleaved)")
    ap.add_argument("--norm-eps", type=float, default=None)
    ap.add_argument("--rope-theta", type=float, default=None)
    ap.add_argument("--max-position", type=int, default=None)
    a = ap.parse_args(argv)
    extra = {k: v for k, v in (("norm_eps", a.norm_eps), ("rope_theta", a.rope_theta), ("max_position", a.max_position)) if v is not None}
    tp = convert_dir(a.model_dir, a.out, a.quant, a.quant_group, a.rope_pairing, extra)
    print(f"wrote {a.out}: params.json + {tp} slice(s), quant={a.quant}")


if __name__ == "__main__":
    main()
