#!/usr/bin/env python3
"""Role-fused launch (pplhip_op_attention_linear) against its two parts launched back to back, on the LLaMA-7B decode shapes:
for each GEMM of the chain (wo, w13+SwiGLU, w2, wqkv) over M rows, with the slice of the other half's attention the schedule
pairs it with (pplhip.cc run_decode_fused).  usage: python profiles/fused_microbench.py [M=512] [KV=512] [NB_TOTAL=512]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.conftest import load_pplhip
m = load_pplhip()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
KV = int(sys.argv[2]) if len(sys.argv) > 2 else 512
NBT = int(sys.argv[3]) if len(sys.argv) > 3 else 512
H, D, hd, inter = 32, 128, 4096, 11008
N_tok = NBT * KV
cache = torch.randint(-127, 128, (2 * H * N_tok * D,), dtype=torch.int8, device="cuda")
kscale = (torch.rand(2 * H * N_tok * D // 8, device="cuda") * 0.02 + 0.01).half()
qkv = torch.randn(NBT, 3 * H * D, device="cuda").half()
att = torch.empty(NBT, H * D, device="cuda", dtype=torch.float16)
seq = torch.arange(NBT + 1, device="cuda", dtype=torch.int64)
sp = torch.full((NBT,), KV - 1, device="cuda", dtype=torch.int64)
ci = torch.arange(NBT, device="cuda", dtype=torch.int64) * KV
v = m.KvView()
v.cache, v.scale, v.max_tokens, v.num_layers, v.kv_heads, v.head_dim = cache.data_ptr(), kscale.data_ptr(), N_tok, 1, H, D
v.quant_bit, v.quant_group, v.layout, v.mode, v.page_size, v.layer = 8, 8, 3, 0, 0, 0
ws = torch.empty(64, device="cuda", dtype=torch.float32)
shapes = [("wo", hd, hd, 0), ("w13", 2 * inter, hd, 1), ("w2", hd, inter, 0), ("wqkv", 3 * hd, hd, 0)]
work = [n * k for _, n, k, _ in shapes]
cuts = np.round(np.cumsum([0] + work) / sum(work) * NBT).astype(int)
L = m.lib()

def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

tot = [0, 0, 0]
for i, (name, N, K, sw) in enumerate(shapes):
    b0, nb = int(cuts[i]), int(cuts[i + 1] - cuts[i])
    x = (torch.randn(M, K, device="cuda") * 0.5).half()
    w = torch.randint(-127, 128, (N, K), dtype=torch.int8, device="cuda")
    sc = (torch.rand(N, device="cuda") * 0.01).half()
    y = torch.empty(M, N, device="cuda", dtype=torch.float16)
    y2 = torch.empty(M, N, device="cuda", dtype=torch.float16)
    a2 = torch.empty_like(att)
    def attn(out=att):
        assert L.pplhip_op_attention(None, qkv.data_ptr(), C.byref(v), seq.data_ptr() + 8 * b0, sp.data_ptr() + 8 * b0, ci.data_ptr() + 8 * b0, 0,
                                     nb, nb, nb, 1, KV, H, 1, ws.data_ptr(), 256, out.data_ptr() + 2 * b0 * H * D) == 0
    def gemm(out=y):
        if sw: assert L.pplhip_op_linear_swiglu(None, x.data_ptr(), w.data_ptr(), sc.data_ptr(), 8, 0, M, N, K, out.data_ptr()) == 0
        else: assert L.pplhip_op_linear(None, x.data_ptr(), w.data_ptr(), sc.data_ptr(), 8, 0, M, N, K, out.data_ptr(), 0) == 0
    def fused():
        assert L.pplhip_op_attention_linear(None, qkv.data_ptr(), C.byref(v), seq.data_ptr() + 8 * b0, sp.data_ptr() + 8 * b0, ci.data_ptr() + 8 * b0,
                                            0, nb, H, a2.data_ptr() + 2 * b0 * H * D, x.data_ptr(), w.data_ptr(), sc.data_ptr(), M, N, K,
                                            y2.data_ptr(), sw) == 0
    ta, tg, tf = timeit(attn), timeit(gemm), timeit(fused)
    same_a = bool((att[b0:b0 + nb] == a2[b0:b0 + nb]).all())
    ncol = N // 2 if sw else N
    dy = (y.view(-1)[:M * ncol].float() - y2.view(-1)[:M * ncol].float()).abs().max().item()
    tot[0] += ta; tot[1] += tg; tot[2] += tf
    print(f"{name:5s} M={M} N={N} K={K} | attention of {nb} requests {ta:7.1f} us | gemm {tg:7.1f} us | sum {ta+tg:7.1f} | fused {tf:7.1f} us"
          f" | attn identical {same_a}, gemm max diff {dy:.2e}")
print(f"slot: attention {tot[0]:.1f} + gemm {tot[1]:.1f} = {tot[0]+tot[1]:.1f} us, fused {tot[2]:.1f} us")
