#!/usr/bin/env python3
"""Exports a HuggingFace LLaMA checkpoint to the model directory this backend loads:

    <out>/params.json                       keys of the reference's params.json (src/common/config.cc:31-148) + this
                                            build's optional keys (norm_eps, rope_theta, max_position, weight_quant_*)
    <out>/model_slice_<rank>/weights.pplhip one container per tensor-parallel rank

It is the counterpart of the reference's `ppl.pmx` export step (docs/llama_guide.md:12-25): merge q/k/v and gate/up,
slice for tensor parallelism (wqkv / w13 / lm_head on the output dim, wo / w2 on the input dim), then
quantise each slice:
    w8a16   per-output-channel symmetric int8:   scale[n] = fp16(max_k |w[n,k]| / 127),  q = clamp(rint(w / scale), -127, 127)
    w4a16   groups of `--quant-group` along K:   scale[n,g] = fp16(max |w| / 7),  q = clamp(rint(w / scale), -8, 7) + 8,
            two nibbles per byte, low nibble = even k
RoPE: HF checkpoints use the half-split pairing this backend implements, so q/k rows are taken as they are.

    python export_hf_llama.py --model-dir /path/to/hf_llama --out /path/to/model --tensor-parallel-size 2 --quant w8a16
"""
import argparse
import glob
import json
import os
import struct
import sys

import numpy as np


def load_state_dict(model_dir):
    """name -> np.ndarray (fp32) from *.safetensors (preferred) or pytorch_model*.bin"""
    sd = {}
    st = sorted(glob.glob(os.path.join(model_dir, "*.safetensors")))
    if st:
        from safetensors import safe_open
        for path in st:
            try:
                with safe_open(path, framework="np") as f:
                    for k in f.keys():
                        sd[k] = f.get_tensor(k)
            except (TypeError, ValueError):   # bf16 checkpoints: numpy has no bfloat16, go through torch
                with safe_open(path, framework="pt") as f:
                    for k in f.keys():
                        sd[k] = f.get_tensor(k).float().numpy()
        return sd
    bins = sorted(glob.glob(os.path.join(model_dir, "pytorch_model*.bin")))
    if not bins:
        raise FileNotFoundError(f"no *.safetensors / pytorch_model*.bin in {model_dir}")
    import torch
    for path in bins:
        for k, v in torch.load(path, map_location="cpu").items():
            sd[k] = v.float().numpy()
    return sd


def write_container(path, tensors):
    """"PPLHIPW1" | u32 count | count x { u32 name_len | name | u64 nbytes | pad to 64 | data }"""
    with open(path, "wb") as f:
        f.write(b"PPLHIPW1")
        f.write(struct.pack("<I", len(tensors)))
        for name, arr in tensors.items():
            arr = np.ascontiguousarray(arr)
            nb = name.encode()
            f.write(struct.pack("<I", len(nb)))
            f.write(nb)
            f.write(struct.pack("<Q", arr.nbytes))
            f.write(b"\0" * ((64 - f.tell() % 64) % 64))
            f.write(arr.tobytes())


def quant_w8(w):
    w = w.astype(np.float32)
    scale = (np.abs(w).max(axis=1) / 127.0).astype(np.float16)
    s32 = scale.astype(np.float32)
    s32[s32 == 0] = 1.0
    q = np.clip(np.rint(w / s32[:, None]), -127, 127).astype(np.int8)
    return q, scale


def quant_w4(w, group):
    w = w.astype(np.float32)
    N, K = w.shape
    if K % group:
        raise ValueError(f"K={K} is not a multiple of the quantisation group {group}")
    g = w.reshape(N, K // group, group)
    scale = (np.abs(g).max(axis=2) / 7.0).astype(np.float16)
    s32 = scale.astype(np.float32)
    s32[s32 == 0] = 1.0
    q = (np.clip(np.rint(g / s32[:, :, None]), -8, 7) + 8).astype(np.uint8).reshape(N, K)
    packed = (q[:, 0::2] | (q[:, 1::2] << 4)).astype(np.uint8)
    return packed, scale


def convert(sd, cfg, tp, quant, group):
    """-> list of per-rank {name: array} in container naming"""
    H, Hkv = cfg["num_attention_heads"], cfg.get("num_key_value_heads", cfg["num_attention_heads"])
    hd, inter, V, L = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"], cfg["num_hidden_layers"]
    D = hd // H
    if H % tp or Hkv % tp or inter % tp or V % tp:
        raise ValueError("heads / kv heads / intermediate size / vocab must be divisible by the tensor-parallel size")
    h, hk, it, vl = H // tp, Hkv // tp, inter // tp, V // tp
    f16 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float16)
    lm_head = sd.get("lm_head.weight", sd["model.embed_tokens.weight"])  # tied embeddings
    slices = []
    for r in range(tp):
        out = {"tok_embeddings.weight": f16(sd["model.embed_tokens.weight"]), "norm.weight": f16(sd["model.norm.weight"]),
               "output.weight": f16(lm_head[r * vl:(r + 1) * vl])}

        def put(name, w):
            if quant == "none":
                out[name + ".weight"] = f16(w)
            elif quant == "w8a16":
                out[name + ".weight"], out[name + ".scale"] = quant_w8(w)
            else:
                out[name + ".weight"], out[name + ".scale"] = quant_w4(w, group)

        for l in range(L):
            p = f"model.layers.{l}."
            out[f"layers.{l}.attention_norm.weight"] = f16(sd[p + "input_layernorm.weight"])
            out[f"layers.{l}.ffn_norm.weight"] = f16(sd[p + "post_attention_layernorm.weight"])
            q, k, v = sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.v_proj.weight"]
            put(f"layers.{l}.attention.wqkv", np.concatenate([q[r * h * D:(r + 1) * h * D], k[r * hk * D:(r + 1) * hk * D],
                                                               v[r * hk * D:(r + 1) * hk * D]], 0))
            put(f"layers.{l}.attention.wo", sd[p + "self_attn.o_proj.weight"][:, r * h * D:(r + 1) * h * D])
            put(f"layers.{l}.feed_forward.w13", np.concatenate([sd[p + "mlp.gate_proj.weight"][r * it:(r + 1) * it],
                                                                  sd[p + "mlp.up_proj.weight"][r * it:(r + 1) * it]], 0))
            put(f"layers.{l}.feed_forward.w2", sd[p + "mlp.down_proj.weight"][:, r * it:(r + 1) * it])
        slices.append(out)
    return slices


def params_json(cfg, args):
    rope_theta = cfg.get("rope_theta")
    if rope_theta is None and isinstance(cfg.get("rope_parameters"), dict):
        rope_theta = cfg["rope_parameters"].get("rope_theta")
    p = {"num_heads": cfg["num_attention_heads"], "num_kv_heads": cfg.get("num_key_value_heads", cfg["num_attention_heads"]),
         "num_layers": cfg["num_hidden_layers"], "hidden_dim": cfg["hidden_size"], "intermediate_dim": cfg["intermediate_size"],
         "vocab_size": cfg["vocab_size"], "cache_quant_bit": args.cache_quant_bit,
         "cache_quant_group": 8 if args.cache_quant_bit == 8 else 1, "cache_layout": args.cache_layout, "cache_mode": args.cache_mode,
         "dynamic_batching": True, "auto_causal": True, "norm_eps": cfg.get("rms_norm_eps", 1e-5),
         "rope_theta": float(rope_theta if rope_theta is not None else 10000.0),
         "max_position": cfg.get("max_position_embeddings", 4096),
         "weight_quant_bit": {"none": 0, "w8a16": 8, "w4a16": 4}[args.quant], "weight_quant_group": args.quant_group}
    if args.cache_mode == 1:
        p["page_size"] = args.page_size
    return p


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--model-dir", required=True, help="HF checkpoint directory (config.json + weights)")
    ap.add_argument("--out", required=True)
    ap.add_argument("--tensor-parallel-size", type=int, default=1)
    ap.add_argument("--quant", choices=["none", "w8a16", "w4a16"], default="w8a16")
    ap.add_argument("--quant-group", type=int, default=128)
    ap.add_argument("--cache-quant-bit", type=int, choices=[0, 8], default=8)
    ap.add_argument("--cache-layout", type=int, choices=[0, 1, 2, 3], default=3)
    ap.add_argument("--cache-mode", type=int, choices=[0, 1], default=0)
    ap.add_argument("--page-size", type=int, default=16)
    args = ap.parse_args(argv)
    cfg = json.load(open(os.path.join(args.model_dir, "config.json")))
    if cfg.get("model_type", "llama") != "llama":
        sys.exit(f"model_type {cfg.get('model_type')} is not LLaMA")
    sd = load_state_dict(args.model_dir)
    slices = convert(sd, cfg, args.tensor_parallel_size, args.quant, args.quant_group)
    os.makedirs(args.out, exist_ok=True)
    json.dump(params_json(cfg, args), open(os.path.join(args.out, "params.json"), "w"), indent=1)
    for r, tensors in enumerate(slices):
        d = os.path.join(args.out, f"model_slice_{r}")
        os.makedirs(d, exist_ok=True)
        write_container(os.path.join(d, "weights.pplhip"), tensors)
    print(f"wrote {args.out}: params.json + {len(slices)} slice(s), quant={args.quant}")


if __name__ == "__main__":
    main()
