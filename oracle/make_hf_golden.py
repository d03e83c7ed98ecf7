#!/usr/bin/env python3
"""Generates tests/golden/hf_tiny_*.npz: golden vectors from an INDEPENDENT oracle.

The reference (OpenPPL/ppl.llm.serving) holds no model arithmetic and no numeric test, so
the CPU restatement in oracle/llama_ref.c is pinned against HuggingFace transformers' LlamaForCausalLM
(fp32, CPU, eager attention) on tiny random LLaMA configs.  This script runs ONLY in the build container
(it needs `transformers`); the fixtures it writes are plain data (weights rounded to fp16, token ids,
logits, greedy continuations) and are committed.  Nothing here is imported by the product.

usage: python oracle/make_hf_golden.py   (writes tests/golden/hf_tiny_mha.npz, hf_tiny_gqa.npz)
"""
import os
import sys

import numpy as np
import torch
from transformers import LlamaConfig, LlamaForCausalLM

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(name, hidden, inter, layers, heads, kv_heads, vocab, seed, prompts, gen_steps):
    torch.manual_seed(seed)
    cfg = LlamaConfig(hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers,
                      num_attention_heads=heads, num_key_value_heads=kv_heads, vocab_size=vocab,
                      rms_norm_eps=1e-5, max_position_embeddings=128, tie_word_embeddings=False,
                      attention_bias=False, mlp_bias=False)
    try:
        cfg.rope_theta = 10000.0
    except Exception:
        pass
    cfg._attn_implementation = "eager"
    model = LlamaForCausalLM(cfg).float().eval()
    # larger-than-default init so logits are well separated; round to fp16 so both sides see equal weights
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 2:
                p.copy_((torch.randn_like(p) * (0.08 if "embed" not in n else 1.0)).half().float())
            else:
                p.copy_((1.0 + 0.1 * torch.randn_like(p)).half().float())
    sd = {k: v.detach().numpy() for k, v in model.state_dict().items()}
    out = {}
    out["tok_embeddings.weight"] = sd["model.embed_tokens.weight"].astype(np.float16)
    out["norm.weight"] = sd["model.norm.weight"].astype(np.float16)
    out["output.weight"] = sd["lm_head.weight"].astype(np.float16)
    for l in range(layers):
        p = f"model.layers.{l}."
        out[f"layers.{l}.attention_norm.weight"] = sd[p + "input_layernorm.weight"].astype(np.float16)
        out[f"layers.{l}.ffn_norm.weight"] = sd[p + "post_attention_layernorm.weight"].astype(np.float16)
        out[f"layers.{l}.attention.wqkv.weight"] = np.concatenate(
            [sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.v_proj.weight"]],
            0).astype(np.float16)
        out[f"layers.{l}.attention.wo.weight"] = sd[p + "self_attn.o_proj.weight"].astype(np.float16)
        out[f"layers.{l}.feed_forward.w13.weight"] = np.concatenate(
            [sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"]], 0).astype(np.float16)
        out[f"layers.{l}.feed_forward.w2.weight"] = sd[p + "mlp.down_proj.weight"].astype(np.float16)

    # greedy generation per prompt, full recompute each step (no HF cache => independent of cache code)
    all_logits = []   # [n_prompts][gen_steps] last-token logits
    all_tokens = []
    hidden_states = None
    for pi, prompt in enumerate(prompts):
        ids = list(prompt)
        lg, tk = [], []
        for s in range(gen_steps):
            with torch.no_grad():
                o = model(torch.tensor([ids]), output_hidden_states=(pi == 0 and s == 0))
            logits = o.logits[0, -1].numpy().astype(np.float32)
            if pi == 0 and s == 0:
                # residual stream after each layer (hidden_states[-1] of HF is post final norm: drop it)
                hidden_states = np.stack([h[0].numpy() for h in o.hidden_states[:-1]]).astype(np.float32)
            lg.append(logits)
            nxt = int(np.argmax(logits))
            tk.append(nxt)
            ids.append(nxt)
        all_logits.append(np.stack(lg))
        all_tokens.append(np.array(tk, dtype=np.int64))
    meta = dict(hidden_dim=hidden, intermediate_dim=inter, num_layers=layers, num_heads=heads, num_kv_heads=kv_heads,
                vocab_size=vocab, norm_eps=1e-5, rope_theta=10000.0, max_position=128)
    path = os.path.join(ROOT, "tests", "golden", f"hf_tiny_{name}.npz")
    np.savez_compressed(
        path, meta_keys=np.array(list(meta.keys())), meta_vals=np.array([float(v) for v in meta.values()]),
        prompts=np.array([np.array(p, dtype=np.int64) for p in prompts], dtype=object),
        logits=np.stack(all_logits), tokens=np.stack(all_tokens), hidden0=hidden_states,
        **{"w:" + k: v for k, v in out.items()})
    print("wrote", path, os.path.getsize(path), "bytes; tokens", [t.tolist() for t in all_tokens])


if __name__ == "__main__":
    build("mha", hidden=128, inter=256, layers=2, heads=4, kv_heads=4, vocab=320, seed=1,
          prompts=[[5, 17, 200, 3, 99], [250, 8, 41], [7, 7, 7, 7, 7, 7, 7, 7, 7, 11, 300, 2, 64]], gen_steps=6)
    build("gqa", hidden=256, inter=384, layers=3, heads=8, kv_heads=2, vocab=512, seed=2,
          prompts=[[1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20], [400, 33]], gen_steps=5)
