#!/usr/bin/env python3
"""Decode attention over a ragged batch (samples_1024-shaped kv lengths) in arrival order vs longest-first: what would a heavy-first
dispatch order of the (head, request) grid buy?   usage: python profiles/probes/decode_ragged_order_probe.py

Measured (round 3): in THIS probe (32 launches back to back, nothing in between) sorting helps, 5.2 -> 5.8 TB/s, in either direction; built into
the runtime (a sorted index array in the step buffer, block y -> request order[y]) and A/B-ed in bench.py's ragged leg (GEMMs between the
attention launches, kv 7 .. 1023, mean 248) it LOSES: 6.0-6.1 -> 5.8 TB/s, twice.  Not adopted; the runtime walks requests in batch order."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.conftest import load_pplhip
m = load_pplhip()
B, H, HKV, D = 1024, 32, 32, 128
rng = np.random.RandomState(1234)
kv = np.clip(rng.lognormal(np.log(150.0), 1.0, size=B), 7, 1023).astype(np.int64)
print("kv mean", kv.mean(), "max", kv.max())
N = int(kv.sum()) + 64
cache = torch.randint(-127, 128, (2 * HKV * N * D,), dtype=torch.int8, device="cuda")
scale = (torch.rand(2 * HKV * N * D // 8, device="cuda") * 0.02 + 0.01).half()
qkv = torch.randn(B, (H + 2 * HKV) * D, device="cuda").half()
out = torch.empty(B, H * D, device="cuda", dtype=torch.float16)
seq = torch.arange(B + 1, device="cuda", dtype=torch.int64)
v = m.KvView()
v.cache, v.scale, v.max_tokens, v.num_layers, v.kv_heads, v.head_dim = cache.data_ptr(), scale.data_ptr(), N, 1, HKV, D
v.quant_bit, v.quant_group, v.layout, v.mode, v.page_size, v.layer = 8, 8, 3, 0, 0, 0
byt = float(kv.sum()) * 2 * HKV * (D + D // 4) + B * H * D * 4
for name, order in (("arrival order", np.arange(B)), ("longest first", np.argsort(-kv, kind="stable")), ("shortest first", np.argsort(kv, kind="stable"))):
    k = kv[order]
    starts = np.concatenate([[0], np.cumsum(kv)])[:-1][order]     # the same slab ranges, visited in another order
    sp = torch.from_numpy((k - 1).astype(np.int64)).cuda()
    ci = torch.from_numpy(starts.astype(np.int64)).cuda()
    call = lambda: m.lib().pplhip_op_attention(None, qkv.data_ptr(), C.byref(v), seq.data_ptr(), sp.data_ptr(), ci.data_ptr(), 0, B, B, B, 1,
                                               int(kv.max()), H, 1, None, 0, out.data_ptr())
    for _ in range(5): assert call() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(32): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 32 * 1e3
    print(f"{name}: {us:.1f} us per launch, {byt / us / 1e3:.0f} GB/s")
