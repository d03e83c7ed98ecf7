#!/usr/bin/env python3
"""W8A16 linears at 4 < M <= 256 on fragment-major weights (csrc/k_gemm_frag.hip) against pplhip_op_linear on the row-major matrix:
agreement of the fp16 outputs (different summation orders, both fp32 accumulation) and HBM-cold times on the 7B layer shapes
(a ring of weight copies larger than the 256 MiB Infinity Cache).
usage: python profiles/frag_microbench.py [M ...]        (PPLHIP_FRAG_BM / _NBW / _SPLITK / _BLOCKS select the launch shape)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.conftest import load_pplhip
m = load_pplhip()
L = m.lib()
shapes = [("wqkv", 12288, 4096, 0), ("wo", 4096, 4096, 0), ("w13", 22016, 4096, 2), ("w2", 4096, 11008, 0)]
MS = [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64, 128, 256]
for M in MS:
    tot = {"row": 0.0, "frag": 0.0}
    tot_b = 0
    line = []
    for name, N, K, epi in shapes:
        g = torch.Generator(device="cuda").manual_seed(N + K + M)
        x = (torch.randn(M, K, device="cuda", generator=g) * 0.5).half()
        w = torch.randint(-127, 128, (N, K), device="cuda", dtype=torch.int8, generator=g)
        sc = (torch.rand(N, device="cuda", generator=g) * 0.001 + 0.0005).half()
        wb = N * K
        NO = N // 2 if epi == 2 else N
        y0 = torch.zeros(M, NO, device="cuda", dtype=torch.float16)
        y1 = torch.full((M, NO), 7.0, device="cuda", dtype=torch.float16)
        copies = max(2, int(600e6 // wb))
        ws = [w.clone() for _ in range(copies)]
        fb = L.pplhip_weight_frag_bytes(N, K)
        wf = [torch.empty(fb, device="cuda", dtype=torch.int8) for _ in range(copies)]
        for i in range(copies):
            assert L.pplhip_weight_pack_frag(None, ws[i].data_ptr(), N, K, wf[i].data_ptr()) == 0
        if epi == 2:
            row = lambda i: L.pplhip_op_linear_swiglu(None, x.data_ptr(), ws[i % copies].data_ptr(), sc.data_ptr(), 8, 128, M, N, K, y0.data_ptr())
        else:
            row = lambda i: L.pplhip_op_linear(None, x.data_ptr(), ws[i % copies].data_ptr(), sc.data_ptr(), 8, 128, M, N, K, y0.data_ptr(), 0)
        frag = lambda i: L.pplhip_op_linear_frag(None, x.data_ptr(), wf[i % copies].data_ptr(), sc.data_ptr(), M, N, K, y1.data_ptr(), epi)
        assert row(0) == 0
        rc = frag(0)
        torch.cuda.synchronize()
        assert rc == 0, rc
        a, b = y0.float(), y1.float()
        d = (a - b).abs()
        bad = int((d > 4e-3 * a.abs() + 4e-3).sum())
        same = float((y0 == y1).float().mean())
        res = {}
        for tag, call in (("row", row), ("frag", frag)):
            for i in range(copies): call(i)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = 2 * copies
                e0.record()
                for i in range(n): call(i)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / n * 1e-3)
            res[tag] = best
            tot[tag] += best
        tot_b += wb
        line.append(f"{name} {res['row'] * 1e6:6.1f} -> {res['frag'] * 1e6:6.1f} us ({wb / res['frag'] / 1e12:4.2f} TB/s, {2.0 * M * N * K / res['frag'] / 1e12:5.0f} TF; eq {same * 100:5.1f} % bad {bad})")
        del ws, wf
    print(f"M={M}: " + " | ".join(line) + f" || layer {tot['row'] * 1e6:7.1f} -> {tot['frag'] * 1e6:7.1f} us = {tot_b / tot['frag'] / 1e12:4.2f} TB/s", flush=True)
