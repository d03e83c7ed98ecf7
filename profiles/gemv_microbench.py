#!/usr/bin/env python3
"""Weight-streaming rate of pplhip_op_linear at M = 1 .. 8 on the 7B layer shapes + lm_head (bytes of weights / time).
usage: python profiles/gemv_microbench.py [wq 8|4|0]      (PPLHIP_GEMV_STREAM_MAX_M=0: the MFMA skinny kernel of round 3)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.conftest import load_pplhip
m = load_pplhip()
WQ = int(sys.argv[1]) if len(sys.argv) > 1 else 8
shapes = [("wqkv", 12288, 4096), ("wo", 4096, 4096), ("w13", 22016, 4096), ("w2", 4096, 11008)]
if os.environ.get("SHAPES"):   # e.g. SHAPES=w13,wo
    shapes = [s for s in shapes if s[0] in os.environ["SHAPES"].split(",")]
MS = [int(a) for a in sys.argv[2:]] or [1, 2, 3, 4, 8]
for M in MS:
    tot_t = tot_b = 0
    line = []
    for name, N, K in shapes:
        x = (torch.randn(M, K, device="cuda") * 0.5).half()
        if WQ == 8:
            w = torch.randint(-127, 128, (N, K), device="cuda", dtype=torch.int8); sc = (torch.rand(N, device="cuda") * 0.001 + 0.0005).half(); wb = N * K
        elif WQ == 4:
            w = torch.randint(0, 256, (N, K // 2), device="cuda", dtype=torch.uint8); sc = (torch.rand(N, K // 128, device="cuda") * 0.01 + 0.005).half(); wb = N * K // 2
        else:
            w = (torch.randn(N, K, device="cuda") * 0.02).half(); sc = torch.zeros(1, device="cuda").half(); wb = N * K * 2
        y = torch.empty(M, N, device="cuda", dtype=torch.float16)
        # a ring of weight copies larger than the 256 MiB Infinity Cache so that every launch streams from HBM
        copies = max(2, int(600e6 // wb))
        ws = [w.clone() for _ in range(copies)]
        call = lambda i: m.lib().pplhip_op_linear(None, x.data_ptr(), ws[i % copies].data_ptr(), sc.data_ptr(), WQ, 128, M, N, K, y.data_ptr(), 0)
        for i in range(copies): call(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 3 * copies
        e0.record()
        for i in range(n): call(i)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / n * 1e-3
        tot_t += t; tot_b += wb
        line.append(f"{name} {t * 1e6:6.1f} us {wb / t / 1e12:4.2f} TB/s")
        del ws
    print(f"M={M} wq={WQ}: " + " | ".join(line) + f" || layer {tot_t * 1e6:7.1f} us = {tot_b / tot_t / 1e12:4.2f} TB/s", flush=True)
