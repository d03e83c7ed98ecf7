#!/usr/bin/env python3
"""Times pplhip_op_linear_i8 (online_i8i8 GEMM, quantiser not included) on the four linear shapes of a LLaMA-2-7B layer.
usage: python profiles/gemm_i8_microbench.py [M]      (PPLHIP_GEMM_I8_VARIANT selects the tile-kernel variant)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.conftest import load_pplhip
m = load_pplhip()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
tot_t = tot_f = 0
for name, N, K in [("wqkv", 12288, 4096), ("wo", 4096, 4096), ("w13", 22016, 4096), ("w2", 4096, 11008)]:
    xq = torch.randint(-127, 128, (M, K), device="cuda", dtype=torch.int8)
    sx = torch.rand(M, device="cuda") * 0.01
    w = torch.randint(-127, 128, (N, K), device="cuda", dtype=torch.int8)
    sc = (torch.rand(N, device="cuda") * 0.001 + 0.0005).half()
    y = torch.empty(M, N, device="cuda", dtype=torch.float16)
    call = lambda: m.lib().pplhip_op_linear_i8(None, xq.data_ptr(), sx.data_ptr(), w.data_ptr(), sc.data_ptr(), M, N, K, y.data_ptr(), 0, 0)
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): call()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20
    fl = 2.0 * M * N * K
    tot_t += t; tot_f += fl
    print(f"{name:5s} M={M} N={N} K={K}: {t*1e3:8.1f} us  {fl/t/1e9:8.1f} TOP/s")
print(f"layer total {tot_t*1e3:.1f} us -> {tot_f/tot_t/1e9:.1f} TOP/s; x32 layers = {tot_t*32:.2f} ms/step")
