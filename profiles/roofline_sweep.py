#!/usr/bin/env python3
"""SURVEY.md D2 "fixed-shape roofline runs": decode attention at B in {1, 64, 256, 512, 1024} x kv_len in {512, 1024, 2048, 4096},
every request at exactly kv_len, pure decode, 64 launches after 8 warm-up launches; int8-g8 KV, cache layout 3, contiguous slots.
Two head geometries: multi-head (LLaMA-2-7B: H = Hkv = 32 -> attn_decode_kernel) and grouped-query (LLaMA-2-70B / TP8 per rank:
H = 8, Hkv = 1 -> attn_decode_gqa_kernel).  The split is the one the runtime's heuristic (pplhip.cc decode_split) picks.
Writes the table as JSON: algorithmic GB per launch, us per launch (HIP events on the launch stream), GB/s, fraction of
8 TB/s.   usage: python profiles/roofline_sweep.py out.json [quick]"""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.conftest import load_pplhip  # noqa: E402

m = load_pplhip()
D = 128


def heuristic_split(B, kv, H, Hkv):
    """pplhip.cc decode_split, mode 1"""
    gqa = 4 <= H // Hkv <= 16
    blocks = B * (Hkv if gqa else H)
    if blocks < 256 and kv >= 512:
        want = (512 + blocks - 1) // blocks
        cap = max(1, kv // 256)
        return max(1, min(want, cap, 32))
    return 1


PAGE = int(os.environ.get("SWEEP_PAGE", "0"))  # SWEEP_PAGE=16: paged cache (shuffled 16-token pages) instead of contiguous slots


def run(B, KV, H, HKV, split, iters=64, warm=8):
    N = B * KV
    cache = torch.randint(-127, 128, (2 * HKV * N * D,), dtype=torch.int8, device="cuda")
    scale = (torch.rand(2 * HKV * N * D // 8, device="cuda") * 0.02 + 0.01).half()
    qkv = torch.randn(B, (H + 2 * HKV) * D, device="cuda").half()
    out = torch.empty(B, H * D, device="cuda", dtype=torch.float16)
    ws = torch.empty(B * H * max(split, 1) * (D + 2) + 64, device="cuda", dtype=torch.float32)
    seq = torch.arange(B + 1, device="cuda", dtype=torch.int64)
    sp = torch.full((B,), KV - 1, device="cuda", dtype=torch.int64)
    ci = torch.arange(B, device="cuda", dtype=torch.int64) * KV
    mp = 0
    if PAGE:
        assert KV % PAGE == 0
        mp = KV // PAGE
        ci = torch.randperm(N // PAGE, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)).to(torch.int64).reshape(B, mp).contiguous()
    v = m.KvView()
    v.cache, v.scale, v.max_tokens, v.num_layers, v.kv_heads, v.head_dim = cache.data_ptr(), scale.data_ptr(), N, 1, HKV, D
    v.quant_bit, v.quant_group, v.layout, v.mode, v.page_size, v.layer = 8, 8, 3, (1 if PAGE else 0), PAGE, 0

    def call():
        return m.lib().pplhip_op_attention(None, qkv.data_ptr(), C.byref(v), seq.data_ptr(), sp.data_ptr(), ci.data_ptr(), mp, B, B, B, 1,
                                           KV, H, split, ws.data_ptr(), ws.numel() * 4, out.data_ptr())
    for _ in range(warm):
        assert call() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    byt = B * KV * 2 * HKV * (D + D // 4) + B * H * D * 4
    del cache, scale
    torch.cuda.empty_cache()
    return {"B": B, "kv_len": KV, "H": H, "Hkv": HKV, "split": split, "us_per_launch": round(us, 2), "algorithmic_GB": round(byt / 1e9, 4),
            "GBps": round(byt / us / 1e3, 1), "frac_of_8TBps": round(byt / us / 1e3 / 8000.0, 4)}


def main():
    out_path = sys.argv[1]
    quick = len(sys.argv) > 2
    rows = []
    for name, H, HKV in (("mha_7b", 32, 32), ("gqa_70b_tp8", 8, 1)):
        for B in ((256, 1024) if quick else (1, 64, 256, 512, 1024)):
            for KV in ((2048,) if quick else (512, 1024, 2048, 4096)):
                r = run(B, KV, H, HKV, heuristic_split(B, KV, H, HKV))
                r["geometry"] = name
                rows.append(r)
                print(json.dumps(r), flush=True)
    json.dump({"what": "decode attention, int8-g8 KV, layout 3, contiguous; 64 launches after 8 warm-up; includes the split-K reduce kernel "
                       "when split > 1 (stream events around the launches)", "peak_GBps": 8000.0, "rows": rows}, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
