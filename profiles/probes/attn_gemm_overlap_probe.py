#!/usr/bin/env python3
"""Does decode attention (HBM-bound) overlap with the layer GEMMs (MFMA-bound) when two half-batches run on two streams?
LLaMA-2-7B layer shapes, W8A16, int8-g8 KV, kv_len 520.  Three schedules over LAYERS layers:
  seq1024 : one stream, batch 1024: attention, then the four GEMMs           (what the runtime does today)
  seq512x2: one stream, two half-batches of 512 one after the other           (cost of halving the batch alone)
  overlap : two streams, half-batch A on one, B on the other, B started half a layer late (its GEMMs run beside A's attention)
usage: python profiles/probes/attn_gemm_overlap_probe.py [LAYERS] [B] [KV] [prio]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.conftest import load_pplhip  # noqa: E402

m = load_pplhip()
LAYERS = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
KV = int(sys.argv[3]) if len(sys.argv) > 3 else 520
PRIO = int(sys.argv[4]) if len(sys.argv) > 4 else 0
H = HKV = 32
D = 128
SHAPES = [("wqkv", 12288, 4096), ("wo", 4096, 4096), ("w13", 22016, 4096), ("w2", 4096, 11008)]


class Half:
    """attention + GEMM operands of `rows` decode rows"""
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
        self.x = {k: (torch.randn(rows, K, device="cuda") * 0.5).half() for k, _, K in SHAPES}
        self.y = {k: torch.empty(rows, N, device="cuda", dtype=torch.float16) for k, N, _ in SHAPES}
        self.ws = torch.empty(64 << 20, device="cuda", dtype=torch.uint8)

    def attn(self, st):
        rc = m.lib().pplhip_op_attention(C.c_void_p(st), self.qkv.data_ptr(), C.byref(self.v), self.seq.data_ptr(), self.sp.data_ptr(),
                                         self.ci.data_ptr(), 0, self.rows, self.rows, self.rows, 1, KV, H, 1, None, 0, self.out.data_ptr())
        assert rc == 0

    def gemms(self, st, W):
        for k, N, K in SHAPES:
            w, sc = W[k]
            rc = m.lib().pplhip_op_linear(C.c_void_p(st), self.x[k].data_ptr(), w.data_ptr(), sc.data_ptr(), 8, 128, self.rows, N, K,
                                          self.y[k].data_ptr(), 0)
            assert rc == 0


W = {k: (torch.randint(-127, 128, (N, K), device="cuda", dtype=torch.int8), (torch.rand(N, device="cuda") * 0.001 + 0.0005).half())
     for k, N, K in SHAPES}
full, ha, hb = Half(B), Half(B // 2), Half(B // 2)
if PRIO:
    s1 = torch.cuda.Stream(priority=0)
    s2 = torch.cuda.Stream(priority=-1)
else:
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def seq1024():
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(LAYERS):
        full.attn(st)
        full.gemms(st, W)


def seq512x2():
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(LAYERS):
        for h in (ha, hb):
            h.attn(st)
            h.gemms(st, W)


def only(which):
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(LAYERS):
        if which == "attn":
            full.attn(st)
        else:
            full.gemms(st, W)


def overlap():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    # A: attn, gemms, attn, gemms ...   B: (A's first attention alone) gemms-then-attn order: B.attn of layer l follows B.gemms of layer l - 1
    ev = torch.cuda.Event()
    for layer in range(LAYERS):
        ha.attn(s1.cuda_stream)
        if layer == 0:
            ev.record(s1)          # B starts when A's first attention is done -> anti-phase from there on
            s2.wait_event(ev)
        ha.gemms(s1.cuda_stream, W)
        hb.attn(s2.cuda_stream)
        hb.gemms(s2.cuda_stream, W)
    cur.wait_stream(s1)
    cur.wait_stream(s2)


def overlap_locked():
    """as overlap, but every B.attention waits for the A.attention of the same layer and every A.attention for B's of the layer before:
    the two attentions never run at the same time"""
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    eb_prev = None
    for layer in range(LAYERS):
        if eb_prev is not None:
            s1.wait_event(eb_prev)
        ha.attn(s1.cuda_stream)
        ea = torch.cuda.Event()
        ea.record(s1)
        ha.gemms(s1.cuda_stream, W)
        s2.wait_event(ea)
        hb.attn(s2.cuda_stream)
        eb_prev = torch.cuda.Event()
        eb_prev.record(s2)
        hb.gemms(s2.cuda_stream, W)
    cur.wait_stream(s1)
    cur.wait_stream(s2)


for f in (seq1024, seq512x2, overlap, overlap_locked):
    f()
torch.cuda.synchronize()
ta, tg = timed(lambda: only("attn")), timed(lambda: only("gemm"))
t0, t1, t2, t3 = timed(seq1024), timed(seq512x2), timed(overlap), timed(overlap_locked)
per = lambda t: t / LAYERS * 1e3
print(f"B={B} kv={KV} layers={LAYERS} prio={PRIO}: per layer  attention alone {per(ta):.0f} us, GEMMs alone {per(tg):.0f} us, "
      f"seq1024 {per(t0):.0f} us, seq512x2 {per(t1):.0f} us, overlap (free-running) {per(t2):.0f} us, overlap (attentions serialised) {per(t3):.0f} us")
