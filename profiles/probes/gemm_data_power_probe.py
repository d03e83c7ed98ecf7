#!/usr/bin/env python3
"""Same kernels, same shapes, different operand DATA: how much of the K-loop rate is the chip's power management (round 4)?
wqkv shape (M 1024, N 12288) at K = 12288 so that the loop dominates; asm-loop kernel and product kernel."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = sys.argv[:1] + ["1024", "0"]
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_asm_probe.py")).read().split("for name, N, K in SHAPES:")[0]
exec(src)
M, N, K = 1024, 12288, 12288
sc = (torch.rand(N, device="cuda") * 0.001 + 0.0005).half()
y = torch.empty(M, N, device="cuda", dtype=torch.float16)
g = torch.Generator(device="cuda").manual_seed(1)
xs = {"x~N(0,0.5)": (torch.randn(M, K, device="cuda", generator=g) * 0.5).half(), "x=0": torch.zeros(M, K, device="cuda").half(),
      "x=1": torch.ones(M, K, device="cuda").half()}
ws = {"w uniform[-127,127]": torch.randint(-127, 128, (N, K), device="cuda", dtype=torch.int8, generator=g),
      "w gauss std 32": (torch.randn(N, K, device="cuda", generator=g) * 32).round().clamp(-127, 127).to(torch.int8),
      "w gauss std 8": (torch.randn(N, K, device="cuda", generator=g) * 8).round().clamp(-127, 127).to(torch.int8),
      "w=0": torch.zeros(N, K, device="cuda", dtype=torch.int8)}
for xn, x in xs.items():
    for wn, w in ws.items():
        if xn != "x~N(0,0.5)" and wn not in ("w=0", "w uniform[-127,127]"):
            continue
        new = lambda: probe.pplhip_probe_linear_w8_asm(None, x.data_ptr(), w.data_ptr(), sc.data_ptr(), M, N, K, y.data_ptr(), 0)
        old = lambda: m.lib().pplhip_op_linear(None, x.data_ptr(), w.data_ptr(), sc.data_ptr(), 8, 128, M, N, K, y.data_ptr(), 0)
        for c in (new, old):
            c(); c()
        tn = min(timeit(new, 30) for _ in range(3)); to = min(timeit(old, 30) for _ in range(3))
        fl = 2.0 * M * N * K
        print(f"{xn:12s} {wn:22s}: asm loop {tn:7.1f} us {fl / tn / 1e6:7.1f} TFLOP/s   product {to:7.1f} us {fl / to / 1e6:7.1f} TFLOP/s", flush=True)
