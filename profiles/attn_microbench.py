#!/usr/bin/env python3
"""Times the decode attention kernel (pplhip_op_attention) for a (B, kv_len, H, Hkv) shape, int8-g8 KV, layout 3.
usage: python profiles/attn_microbench.py B KV H HKV [split]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.conftest import load_pplhip
m = load_pplhip()
B, KV, H, HKV = [int(x) for x in sys.argv[1:5]]
split = int(sys.argv[5]) if len(sys.argv) > 5 else 1
D, L = 128, 1
N = B * KV
cache = torch.randint(-127, 128, (L * 2 * HKV * N * D,), dtype=torch.int8, device="cuda")
scale = (torch.rand(L * 2 * HKV * N * D // 8, device="cuda") * 0.02 + 0.01).half()
qkv = torch.randn(B, (H + 2 * HKV) * D, device="cuda").half()
out = torch.empty(B, H * D, device="cuda", dtype=torch.float16)
ws = torch.empty(B * H * max(split, 1) * (D + 2) + 64, device="cuda", dtype=torch.float32)
seq = torch.arange(B + 1, device="cuda", dtype=torch.int64)
sp = torch.full((B,), KV - 1, device="cuda", dtype=torch.int64)
ci = torch.arange(B, device="cuda", dtype=torch.int64) * KV
v = m.KvView()
v.cache, v.scale, v.max_tokens, v.num_layers, v.kv_heads, v.head_dim = cache.data_ptr(), scale.data_ptr(), N, L, HKV, D
v.quant_bit, v.quant_group, v.layout, v.mode, v.page_size, v.layer = 8, 8, 3, 0, 0, 0
call = lambda: m.lib().pplhip_op_attention(None, qkv.data_ptr(), C.byref(v), seq.data_ptr(), sp.data_ptr(), ci.data_ptr(), 0, B, B, B, 1, KV, H,
                                           split, ws.data_ptr(), ws.numel() * 4, out.data_ptr())
for _ in range(3): assert call() == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): call()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10
byt = B * KV * 2 * HKV * (D + D // 4) + B * H * D * 4
print(f"B={B} kv={KV} H={H} Hkv={HKV} split={split}: {t*1e3:.1f} us, algorithmic {byt/1e9:.3f} GB -> {byt/t/1e6:.0f} GB/s")
