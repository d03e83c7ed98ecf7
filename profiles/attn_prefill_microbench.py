#!/usr/bin/env python3
"""Times the prefill attention kernel (pplhip_op_attention, decoding_batches = 0) for R requests of S new tokens each over a
cache of P earlier tokens (int8-g8 KV, layout 3), contiguous or paged.
usage: python profiles/attn_prefill_microbench.py R S [P] [mode 0|1] [H] [HKV]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.conftest import load_pplhip
m = load_pplhip()
R, S = int(sys.argv[1]), int(sys.argv[2])
P = int(sys.argv[3]) if len(sys.argv) > 3 else 0
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0
H = int(sys.argv[5]) if len(sys.argv) > 5 else 32
HKV = int(sys.argv[6]) if len(sys.argv) > 6 else H
D, L, PG = 128, 1, 16
per = (P + S + PG - 1) // PG * PG
N = R * per
cache = torch.randint(-127, 128, (L * 2 * HKV * N * D,), dtype=torch.int8, device="cuda")
scale = (torch.rand(L * 2 * HKV * N * D // 8, device="cuda") * 0.02 + 0.01).half()
qkv = torch.randn(R * S, (H + 2 * HKV) * D, device="cuda").half()
out = torch.empty(R * S, H * D, device="cuda", dtype=torch.float16)
seq = (torch.arange(R + 1, device="cuda", dtype=torch.int64) * S)
sp = torch.full((R,), P, device="cuda", dtype=torch.int64)
if mode == 0:
    ci = torch.arange(R, device="cuda", dtype=torch.int64) * per
    mp = 0
else:
    mp = per // PG
    ci = torch.from_numpy(np.random.RandomState(0).permutation(N // PG).astype(np.int64).reshape(R, mp)).cuda()
v = m.KvView()
v.cache, v.scale, v.max_tokens, v.num_layers, v.kv_heads, v.head_dim = cache.data_ptr(), scale.data_ptr(), N, L, HKV, D
v.quant_bit, v.quant_group, v.layout, v.mode, v.page_size, v.layer = 8, 8, 3, mode, PG if mode else 0, 0
# MB_WS=1: hand the operator a workspace (enables the split-KV form for short suffixes behind long caches)
ws = torch.empty(R * S * H * min(32, max(2, 512 // max(1, ((S + 127) // 128) * R * H) + 1)) * (D + 2), device="cuda", dtype=torch.float32) if os.environ.get("MB_WS") and S < 1024 else None
call = lambda: m.lib().pplhip_op_attention(None, qkv.data_ptr(), C.byref(v), seq.data_ptr(), sp.data_ptr(), ci.data_ptr(), mp, R, R * S, 0, S,
                                           P + S, H, 1, ws.data_ptr() if ws is not None else None, ws.numel() * 4 if ws is not None else 0,
                                           out.data_ptr())
for _ in range(3): assert call() == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): call()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10
fl = 4.0 * H * D * R * sum(P + i + 1 for i in range(S))
print(f"R={R} S={S} P={P} mode={mode} H={H} Hkv={HKV}: {t*1e3:.1f} us, {fl/t/1e9:.1f} TFLOP/s (causal flops)")
