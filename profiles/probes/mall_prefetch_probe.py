#!/usr/bin/env python3
"""Round 4: does a decode-step linear run faster when its weights were read once just before (resident in the 256 MiB Infinity Cache, not
in L2) than when they stream from HBM?  -> is a prefetch branch beside the step's short kernels (norms, RoPE, attention at small batch) worth building?
For each 7B layer shape and M: time pplhip_op_linear alone (events around the one launch) on a ring of weight copies > 256 MiB, (a) cold,
(b) after a kernel that read the same copy (torch sum over the int8 bytes as int32 words), (c) after the same kernel read ANOTHER 64 MiB on top.
usage: python profiles/probes/mall_prefetch_probe.py [M ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.conftest import load_pplhip
m = load_pplhip()
shapes = [("wqkv", 12288, 4096), ("wo", 4096, 4096), ("w13", 22016, 4096), ("w2", 4096, 11008)]
MS = [int(a) for a in sys.argv[1:]] or [1, 8, 64]
other = torch.randint(-127, 128, (64 << 20,), device="cuda", dtype=torch.int8)
for M in MS:
    for name, N, K in shapes:
        x = (torch.randn(M, K, device="cuda") * 0.5).half()
        w = torch.randint(-127, 128, (N, K), device="cuda", dtype=torch.int8)
        sc = (torch.rand(N, device="cuda") * 0.001 + 0.0005).half()
        y = torch.empty(M, N, device="cuda", dtype=torch.float16)
        copies = max(3, int(700e6 // (N * K)))
        ws = [w.clone() for _ in range(copies)]
        call = lambda i: m.lib().pplhip_op_linear(None, x.data_ptr(), ws[i % copies].data_ptr(), sc.data_ptr(), 8, 128, M, N, K, y.data_ptr(), 0)
        res = {}
        for mode in ("cold", "touched", "touched+64MiB"):
            ts = []
            for it in range(3 * copies):
                i = it % copies
                if mode != "cold":
                    ws[i].view(torch.int32).sum()
                    if mode == "touched+64MiB": other.view(torch.int32).sum()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); call(i); e1.record()
                torch.cuda.synchronize()
                if it >= copies: ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            res[mode] = ts[len(ts) // 2]
        print(f"M={M} {name:5s} {N * K / 1e6:5.1f} MB: " + "  ".join(f"{k} {v:6.1f} us ({N * K / v / 1e6:4.2f} TB/s)" for k, v in res.items()), flush=True)
        del ws
