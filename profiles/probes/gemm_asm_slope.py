#!/usr/bin/env python3
"""time per K tile of the asm-loop GEMM (and its diagnosis builds) from the slope between K = 4096 and K = 12288 at N = 12288, M = 1024
(256 blocks: one per CU) -- launch, prologue and epilogue cancel out"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = sys.argv[:1] + ["1024", "0"]
import importlib.util
spec = importlib.util.spec_from_file_location("gp", os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_asm_probe.py"))
# reuse the loaders without running the shape loop
src = open(spec.origin).read().split("for name, N, K in SHAPES:")[0]
exec(src)
M, N = 1024, 12288
res = {}
for K in (4096, 12288):
    x = (torch.randn(M, K, device="cuda") * 0.5).half()
    w = torch.randint(-127, 128, (N, K), device="cuda", dtype=torch.int8)
    sc = (torch.rand(N, device="cuda") * 0.001 + 0.0005).half()
    y = torch.empty(M, N, device="cuda", dtype=torch.float16)
    libs = {"base": probe, **ablations}
    calls = {n: (lambda l=l: l.pplhip_probe_linear_w8_asm(None, x.data_ptr(), w.data_ptr(), sc.data_ptr(), M, N, K, y.data_ptr(), 0)) for n, l in libs.items()}
    calls["product"] = lambda: m.lib().pplhip_op_linear(None, x.data_ptr(), w.data_ptr(), sc.data_ptr(), 8, 128, M, N, K, y.data_ptr(), 0)
    for n, c in calls.items():
        c(); c()
        res[(n, K)] = min(timeit(c) for _ in range(5))
for n in calls:
    t1, t2 = res[(n, 4096)], res[(n, 12288)]
    per_tile = (t2 - t1) / 128.0
    print(f"{n:8s} K=4096 {t1:7.1f} us  K=12288 {t2:7.1f} us  -> {per_tile * 1e3:7.1f} ns per K tile ({per_tile * 1e3 / 48:5.2f} ns per MFMA), fixed part {t1 - 64 * per_tile:6.1f} us,"
          f" loop rate {2.0 * M * N * 64 / per_tile / 1e6:7.1f} TFLOP/s")
