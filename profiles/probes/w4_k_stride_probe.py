#!/usr/bin/env python3
"""Round 5: does the power-of-two row stride of x (K = 8192 -> 16 KiB) cost the M = 256 W4 GEMMs L2 channel conflicts?
Times pplhip_op_linear (W4A16, M = 256, N = 7168) at K = 8192 and at neighbouring K that are multiples of 128 but not of 2048."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.conftest import load_pplhip
m = load_pplhip()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for N, K in [(7168, 8192), (7168, 8320), (7168, 8064), (7168, 7936), (7168, 8192), (8192, 3584), (8192, 3712), (8192, 4096)]:
    x = (torch.randn(M, K, device="cuda") * 0.5).half()
    w = torch.randint(0, 256, (N, K // 2), device="cuda", dtype=torch.uint8)
    sc = (torch.rand(N, K // 128, device="cuda") * 0.01 + 0.005).half()
    y = torch.empty(M, N, device="cuda", dtype=torch.float16)
    call = lambda: m.lib().pplhip_op_linear(None, x.data_ptr(), w.data_ptr(), sc.data_ptr(), 4, 128, M, N, K, y.data_ptr(), 0)
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): call()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20
    print(f"M={M} N={N} K={K}: {t*1e3:8.1f} us  {2.0*M*N*K/t/1e9:8.1f} TFLOP/s  ({t*1e3/(K/128):.3f} us per 128-deep step)")
