#!/usr/bin/env python3
"""decode attention (B = 1024, H = 32, int8-g8 KV) time per KV token across kv lengths around a multiple of the block's 128-token
iteration, ONE process, interleaved rounds: what does the nearly empty last iteration of kv = 513 .. 539 cost?"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.conftest import load_pplhip
m = load_pplhip()
B, H, D = 1024, 32, 128
KVS = [int(a) for a in sys.argv[1:]] or [384, 500, 512, 513, 520, 539, 544, 576, 640]
KMAX = max(KVS)
N = B * KMAX
cache = torch.randint(-127, 128, (2 * H * N * D,), dtype=torch.int8, device="cuda")
scale = (torch.rand(2 * H * N * D // 8, device="cuda") * 0.02 + 0.01).half()
qkv = torch.randn(B, 3 * H * D, device="cuda").half()
out = torch.empty(B, H * D, device="cuda", dtype=torch.float16)
seq = torch.arange(B + 1, device="cuda", dtype=torch.int64)
ci = torch.arange(B, device="cuda", dtype=torch.int64) * KMAX
v = m.KvView()
v.cache, v.scale, v.max_tokens, v.num_layers, v.kv_heads, v.head_dim = cache.data_ptr(), scale.data_ptr(), N, 1, H, D
v.quant_bit, v.quant_group, v.layout, v.mode, v.page_size, v.layer = 8, 8, 3, 0, 0, 0
sps = {kv: torch.full((B,), kv - 1, device="cuda", dtype=torch.int64) for kv in KVS}
def run(kv, n=8):
    sp = sps[kv]
    call = lambda: m.lib().pplhip_op_attention(None, qkv.data_ptr(), C.byref(v), seq.data_ptr(), sp.data_ptr(), ci.data_ptr(), 0, B, B, B, 1, kv, H, 1, None, 0, out.data_ptr())
    call(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
best = {kv: 1e9 for kv in KVS}
for r in range(4):
    for kv in KVS:
        best[kv] = min(best[kv], run(kv))
for kv in KVS:
    by = B * kv * 2 * H * (D + D // 4) + B * H * D * 4
    print(f"kv {kv:5d}: {best[kv]:8.1f} us  {best[kv] / kv * 1e3 / 1.0:7.1f} ns per kv position (batch 1024)  {by / best[kv] / 1e3:7.1f} GB/s")
