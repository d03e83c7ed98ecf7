#!/usr/bin/env python3
"""Probe: CU-masked streams (hipExtStreamCreateWithCUMask).  How do the decode attention (HBM-bound) and the layer GEMMs
(MFMA-bound) behave on a subset of the CUs, and do they overlap when each gets its own subset?
usage: python profiles/overlap_probe3.py"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.conftest import load_pplhip
m = load_pplhip()
hip = C.CDLL("libamdhip64.so")

def masked_stream(bits):
    """bits: iterable of CU indices (0..255) enabled"""
    words = [0] * 8
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    arr = (C.c_uint32 * 8)(*words)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, arr)
    assert rc == 0, rc
    return s

B, KV, H, D, L = 512, 512, 32, 128, 1
M = 512
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
gem = []
for N, K in [(12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008)]:
    x = (torch.randn(M, K, device="cuda") * 0.5).half()
    w = torch.randint(-127, 128, (N, K), device="cuda", dtype=torch.int8)
    sc = (torch.rand(N, device="cuda") * 0.001 + 0.0005).half()
    y = torch.empty(M, N, device="cuda", dtype=torch.float16)
    gem.append((x, w, sc, y, N, K))

def attn(s):
    assert m.lib().pplhip_op_attention(s, qkv.data_ptr(), C.byref(v), seq.data_ptr(), sp.data_ptr(), ci.data_ptr(), 0, B, B, B, 1, KV, H, 1,
                                       None, 0, out.data_ptr()) == 0
def gemms(s):
    for x, w, sc, y, N, K in gem:
        assert m.lib().pplhip_op_linear(s, x.data_ptr(), w.data_ptr(), sc.data_ptr(), 8, 128, M, N, K, y.data_ptr(), 0) == 0

ev = [C.c_void_p(), C.c_void_p()]
for e in ev: assert hip.hipEventCreate(C.byref(e)) == 0
def timed(fns_streams, reps=20):
    """fns_streams: list of (fn, stream); all launched per rep; time = wall on device from first start to last end"""
    for fn, s in fns_streams:
        for _ in range(3): fn(s)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(reps):
        for fn, s in fns_streams: fn(s)
    for fn, s in fns_streams: assert hip.hipStreamSynchronize(s) == 0
    return (time.perf_counter() - t0) / reps * 1e6

full = masked_stream(range(256))
attn(full); gemms(full); torch.cuda.synchronize()
print(f"full mask: attention {timed([(attn, full)]):.0f} us, gemms {timed([(gemms, full)]):.0f} us")
for name, sel in [("first 128 bits", range(128)), ("even bits", range(0, 256, 2)), ("first 64", range(64)), ("bits = 0 mod 4", range(0, 256, 4)),
                  ("first 192", range(192)), ("first 160", range(160)), ("first 96", range(96))]:
    s = masked_stream(sel)
    print(f"{name:16s} ({len(list(sel))} CUs): attention {timed([(attn, s)]):.0f} us, gemms {timed([(gemms, s)]):.0f} us")
for na, label in [(128, "128/128"), (96, "96/160"), (64, "64/192"), (160, "160/96")]:
    for kind in ("contiguous", "interleaved"):
        if kind == "contiguous":
            a_bits, g_bits = range(na), range(na, 256)
        else:
            period = 8
            k = na * period // 256
            a_bits = [i for i in range(256) if (i % period) < k]
            g_bits = [i for i in range(256) if (i % period) >= k]
        sa, sg = masked_stream(a_bits), masked_stream(g_bits)
        ta, tg = timed([(attn, sa)]), timed([(gemms, sg)])
        tb = timed([(attn, sa), (gemms, sg)])
        print(f"split attn/gemm {label} {kind:11s}: attention {ta:.0f} us, gemms {tg:.0f} us, concurrent {tb:.0f} us")
