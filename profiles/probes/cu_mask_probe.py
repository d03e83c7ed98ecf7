#!/usr/bin/env python3
"""How much HBM bandwidth does decode attention reach on a SUBSET of the CUs (hipExtStreamCreateWithCUMask), and how fast are the layer
GEMMs on the rest?  -> can a step run attention of one half-batch beside the GEMMs of the other on disjoint CU sets?
usage: python profiles/probes/cu_mask_probe.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.conftest import load_pplhip  # noqa: E402

m = load_pplhip()
hip = C.CDLL("libamdhip64.so")
H = HKV = 32
D = 128
KV = 520
SHAPES = [("wqkv", 12288, 4096), ("wo", 4096, 4096), ("w13", 22016, 4096), ("w2", 4096, 11008)]


def masked_stream(bits):
    """bits: iterable of CU indices (0..255) that are enabled"""
    words = (C.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    assert rc == 0, rc
    return st


class Att:
    def __init__(self, rows):
        self.rows = rows
        n = rows * KV
        self.cache = torch.randint(-127, 128, (2 * HKV * n * D,), dtype=torch.int8, device="cuda")
        self.scale = (torch.rand(2 * HKV * n * D // 8, device="cuda") * 0.02 + 0.01).half()
        self.qkv = torch.randn(rows, (H + 2 * HKV) * D, device="cuda").half()
        self.out = torch.empty(rows, H * D, device="cuda", dtype=torch.float16)
        self.seq = torch.arange(rows + 1, device="cuda", dtype=torch.int64)
        self.sp = torch.full((rows,), KV - 1, device="cuda", dtype=torch.int64)
        self.ci = torch.arange(rows, device="cuda", dtype=torch.int64) * KV
        v = m.KvView()
        v.cache, v.scale, v.max_tokens, v.num_layers, v.kv_heads, v.head_dim = self.cache.data_ptr(), self.scale.data_ptr(), n, 1, HKV, D
        v.quant_bit, v.quant_group, v.layout, v.mode, v.page_size, v.layer = 8, 8, 3, 0, 0, 0
        self.v = v
        self.bytes = rows * KV * 2 * HKV * (D + D // 4)

    def run(self, st):
        assert m.lib().pplhip_op_attention(st, self.qkv.data_ptr(), C.byref(self.v), self.seq.data_ptr(), self.sp.data_ptr(),
                                           self.ci.data_ptr(), 0, self.rows, self.rows, self.rows, 1, KV, H, 1, None, 0, self.out.data_ptr()) == 0


class Gemm:
    def __init__(self, rows):
        self.rows = rows
        self.x = {k: (torch.randn(rows, K, device="cuda") * 0.5).half() for k, _, K in SHAPES}
        self.y = {k: torch.empty(rows, N, device="cuda", dtype=torch.float16) for k, N, _ in SHAPES}
        self.W = {k: (torch.randint(-127, 128, (N, K), device="cuda", dtype=torch.int8), (torch.rand(N, device="cuda") * 0.001 + 0.0005).half())
                  for k, N, K in SHAPES}
        self.flops = sum(2.0 * rows * N * K for _, N, K in SHAPES)

    def run(self, st):
        for k, N, K in SHAPES:
            w, sc = self.W[k]
            assert m.lib().pplhip_op_linear(st, self.x[k].data_ptr(), w.data_ptr(), sc.data_ptr(), 8, 128, self.rows, N, K, self.y[k].data_ptr(), 0) == 0


def wall(fns_streams, iters=10):
    """fns_streams: list of (callable(stream), stream); all launched interleaved, wall time until every stream is idle"""
    for f, st in fns_streams:
        f(st)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(iters):
        for f, st in fns_streams:
            f(st)
    for _, st in fns_streams:
        hip.hipStreamSynchronize(st)
    return (time.perf_counter() - t0) / iters * 1e6


att_full, att_half = Att(1024), Att(512)
g_full, g_half = Gemm(1024), Gemm(512)
all_cus = masked_stream(range(256))
print(f"all 256 CUs: attention(1024) {wall([(att_full.run, all_cus)]):.0f} us, attention(512) {wall([(att_half.run, all_cus)]):.0f} us, "
      f"GEMMs(1024) {wall([(g_full.run, all_cus)]):.0f} us, GEMMs(512) {wall([(g_half.run, all_cus)]):.0f} us")
for name, sel in (("low half (bits 0..127)", lambda i: i < 128), ("even bits", lambda i: i % 2 == 0), ("bits with (i//4)%2==0", lambda i: (i // 4) % 2 == 0),
                  ("bits with (i//8)%2==0", lambda i: (i // 8) % 2 == 0), ("bits with (i//16)%2==0", lambda i: (i // 16) % 2 == 0),
                  ("bits with (i//32)%2==0 (words 0,2,4,6)", lambda i: (i // 32) % 2 == 0), ("3/8: i%8<3", lambda i: i % 8 < 3), ("5/8: i%8<5", lambda i: i % 8 < 5)):
    a = [i for i in range(256) if sel(i)]
    b = [i for i in range(256) if not sel(i)]
    sa, sb = masked_stream(a), masked_stream(b)
    ta = wall([(att_half.run, sa)])
    tg = wall([(g_half.run, sb)])
    both = wall([(att_half.run, sa), (g_half.run, sb)])
    print(f"{name}: {len(a)} CUs attention(512) {ta:.0f} us = {att_half.bytes / ta / 1e6:.2f} TB/s | {len(b)} CUs GEMMs(512) {tg:.0f} us = "
          f"{g_half.flops / tg / 1e6:.0f} TFLOP/s | both at once {both:.0f} us")
