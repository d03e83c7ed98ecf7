#!/usr/bin/env python3
"""Probe: does ONE decode-attention launch (HBM-bound, half batch 512 x kv 512, 7B heads) overlap with the layer GEMMs of
the other half batch (MFMA-bound, M = 512) when they are launched on two HIP streams at the same time?
Prints attention alone, GEMMs alone, both concurrently."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.conftest import load_pplhip
m = load_pplhip()
B, KV, H, D, L = 512, 512, 32, 128, 1
M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N_tok = B * KV
cache = torch.randint(-127, 128, (L * 2 * H * N_tok * D,), dtype=torch.int8, device="cuda")
scale = (torch.rand(L * 2 * H * N_tok * D // 8, device="cuda") * 0.02 + 0.01).half()
qkv = torch.randn(B, 3 * H * D, device="cuda").half()
out = torch.empty(B, H * D, device="cuda", dtype=torch.float16)
seq = torch.arange(B + 1, device="cuda", dtype=torch.int64)
sp = torch.full((B,), KV - 1, device="cuda", dtype=torch.int64)
ci = torch.arange(B, device="cuda", dtype=torch.int64) * KV
v = m.KvView()
v.cache, v.scale, v.max_tokens, v.num_layers, v.kv_heads, v.head_dim = cache.data_ptr(), scale.data_ptr(), N_tok, L, H, D
v.quant_bit, v.quant_group, v.layout, v.mode, v.page_size, v.layer = 8, 8, 3, 0, 0, 0
shapes = [(12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008)]
gem = []
for N, K in shapes:
    x = (torch.randn(M, K, device="cuda") * 0.5).half()
    w = torch.randint(-127, 128, (N, K), device="cuda", dtype=torch.int8)
    sc = (torch.rand(N, device="cuda") * 0.001 + 0.0005).half()
    y = torch.empty(M, N, device="cuda", dtype=torch.float16)
    gem.append((x, w, sc, y, N, K))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def attn(stream):
    assert m.lib().pplhip_op_attention(C.c_void_p(stream.cuda_stream), qkv.data_ptr(), C.byref(v), seq.data_ptr(), sp.data_ptr(), ci.data_ptr(), 0,
                                       B, B, B, 1, KV, H, 1, None, 0, out.data_ptr()) == 0

def gemms(stream):
    for x, w, sc, y, N, K in gem:
        assert m.lib().pplhip_op_linear(C.c_void_p(stream.cuda_stream), x.data_ptr(), w.data_ptr(), sc.data_ptr(), 8, 128, M, N, K, y.data_ptr(), 0) == 0

def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream())
    s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
    for _ in range(reps): fn()
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    e1.record(torch.cuda.current_stream())
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

class _Null:
    cuda_stream = 0
tn = timed(lambda: attn(_Null()))
print(f"attention on the null stream {tn:.1f} us")
ta = timed(lambda: attn(s1))
tg = timed(lambda: gemms(s2))
tb = timed(lambda: (attn(s1), gemms(s2)))
tb2 = timed(lambda: (gemms(s2), attn(s1)))
print(f"attention alone {ta:.1f} us | layer GEMMs (M={M}) alone {tg:.1f} us | sum {ta+tg:.1f} us | concurrent {tb:.1f} us (attn first) {tb2:.1f} us (gemm first)")
