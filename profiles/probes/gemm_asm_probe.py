#!/usr/bin/env python3
"""Round 4: the hand-scheduled W8A16 K loop (csrc/k_gemm_asm.hip built on its own as profiles/probes/libgemm_asm_probe.so) against the
product's pplhip_op_linear on the same operands: bit-level agreement of the fp16 outputs (the product kernels are held against the oracle by
tests/test_gpu_ops.py; both accumulate in fp32, in different orders) and interleaved timings.
usage: python profiles/probes/gemm_asm_probe.py [M] [rounds]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.conftest import load_pplhip  # noqa: E402

m = load_pplhip()
import glob
vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int32


def load(path):
    lib = C.CDLL(path)
    lib.pplhip_probe_linear_w8_asm.argtypes = [vp, vp, vp, vp, i64, i32, i32, vp, i32]
    lib.pplhip_probe_linear_w8_asm.restype = C.c_int
    return lib


probe = load(os.path.join(ROOT, "profiles", "probes", "libgemm_asm_probe.so"))
# diagnosis builds (wrong results, same instruction stream otherwise): build_gemm_asm_probe.sh
ablations = {os.path.basename(p)[len("libgemm_asm_probe_"):-3]: load(p)
             for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "probes", "libgemm_asm_probe_abl*.so")))}
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
SHAPES = [("wqkv", 12288, 4096), ("w13", 22016, 4096), ("wo", 4096, 4096), ("w2", 4096, 11008), ("odd", 1000, 192), ("k64", 768, 64)]


def timeit(call, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, N, K in SHAPES:
    Mx = M if name not in ("odd", "k64") else 333
    g = torch.Generator(device="cuda").manual_seed(N + K)
    x = (torch.randn(Mx, K, device="cuda", generator=g) * 0.5).half()
    w = torch.randint(-127, 128, (N, K), device="cuda", dtype=torch.int8, generator=g)
    sc = (torch.rand(N, device="cuda", generator=g) * 0.001 + 0.0005).half()
    y0 = torch.zeros(Mx, N, device="cuda", dtype=torch.float16)
    y1 = torch.full((Mx, N), 7.0, device="cuda", dtype=torch.float16)
    old = lambda: m.lib().pplhip_op_linear(None, x.data_ptr(), w.data_ptr(), sc.data_ptr(), 8, 128, Mx, N, K, y0.data_ptr(), 0)
    new = lambda: probe.pplhip_probe_linear_w8_asm(None, x.data_ptr(), w.data_ptr(), sc.data_ptr(), Mx, N, K, y1.data_ptr(), 0)
    assert old() == 0
    rc = new()
    torch.cuda.synchronize()
    assert rc == 0, rc
    a, b = y0.float(), y1.float()
    ref = x.float() @ w.float().t() * sc.float()[None, :]           # fp32 reference on the device (another summation order)
    d = (a - b).abs()
    same = float((y0 == y1).float().mean())
    e_old, e_new = float((a - ref).abs().max()), float((b - ref).abs().max())
    print(f"{name:5s} M={Mx} N={N} K={K}: bit-equal {same * 100:.3f} %  max|new-old| {float(d.max()):.3e}  |old-ref| {e_old:.3e}  |new-ref| {e_new:.3e}"
          f"  |y|max {float(a.abs().max()):.3f}", flush=True)
    bad = (d > 2e-3 * a.abs() + 2e-3).nonzero()
    if len(bad):
        print("   MISMATCH rows/cols (first 8):", bad[:8].tolist(), " of ", len(bad))
        continue
    if name in ("w13", "odd"):   # the fused SwiGLU epilogue against the product's
        ys0 = torch.zeros(Mx, N // 2, device="cuda", dtype=torch.float16)
        ys1 = torch.full((Mx, N // 2), 7.0, device="cuda", dtype=torch.float16)
        assert m.lib().pplhip_op_linear_swiglu(None, x.data_ptr(), w.data_ptr(), sc.data_ptr(), 8, 128, Mx, N, K, ys0.data_ptr()) == 0
        assert probe.pplhip_probe_linear_w8_asm(None, x.data_ptr(), w.data_ptr(), sc.data_ptr(), Mx, N, K, ys1.data_ptr(), 2) == 0
        torch.cuda.synchronize()
        ds = (ys0.float() - ys1.float()).abs()
        print(f"      swiglu: bit-equal {float((ys0 == ys1).float().mean()) * 100:.3f} %  max|new-old| {float(ds.max()):.3e}  |y|max {float(ys0.float().abs().max()):.3f}"
              f"  bad {int((ds > 4e-3 * ys0.float().abs() + 4e-3).sum())}", flush=True)
    if name in ("odd", "k64"):
        continue
    for _ in range(3):
        old(); new()
    to, tn = [], []
    for _ in range(ROUNDS):
        to.append(timeit(old)); tn.append(timeit(new))
    fl = 2.0 * Mx * N * K
    if name in ("wqkv", "w13"):
        y2 = torch.empty_like(y1)
        for an, lib in ablations.items():
            ab = lambda: lib.pplhip_probe_linear_w8_asm(None, x.data_ptr(), w.data_ptr(), sc.data_ptr(), Mx, N, K, y2.data_ptr(), 0)
            ab(); ab()
            ta = [timeit(ab) for _ in range(3)]
            print(f"      {an:6s} {min(ta):7.1f} us  {fl / min(ta) / 1e6:7.1f} TFLOP/s", flush=True)
    print(f"      product {min(to):7.1f} us (median {sorted(to)[len(to) // 2]:7.1f})  {fl / min(to) / 1e6:7.1f} TFLOP/s   |   asm loop {min(tn):7.1f} us "
          f"(median {sorted(tn)[len(tn) // 2]:7.1f})  {fl / min(tn) / 1e6:7.1f} TFLOP/s", flush=True)
