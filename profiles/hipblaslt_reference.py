#!/usr/bin/env python3
"""Context for the GEMM roofline numbers: what the vendor's tuned library (hipBLASLt behind torch.matmul, fp16 x fp16 -> fp16, random
operands) reaches on the layer shapes of LLaMA-2-7B at the batch sizes of the benchmark.  A measurement aid only -- nothing in the
product calls a library GEMM.  usage: python profiles/hipblaslt_reference.py"""
import torch
SHAPES = [("wqkv", 12288, 4096), ("wo", 4096, 4096), ("w13", 22016, 4096), ("w2", 4096, 11008)]
TP4 = [("wqkv", 3072, 4096), ("wo", 4096, 1024), ("w13", 5504, 4096), ("w2", 4096, 2752)]   # one rank's slice at tensor-parallel 4 / 8 (round 6)
TP8 = [("wqkv", 1536, 4096), ("wo", 4096, 512), ("w13", 2752, 4096), ("w2", 4096, 1408)]
for M, shapes, label in ((1024, SHAPES, "7b"), (8192, SHAPES, "7b"), (1024, TP4, "7b-tp4"), (1024, TP8, "7b-tp8")):
    tot_t = tot_f = 0.0
    print(f"# {label} M={M}")
    for name, N, K in shapes:
        a = torch.randn(M, K, device="cuda", dtype=torch.float16)
        b = torch.randn(N, K, device="cuda", dtype=torch.float16)
        for _ in range(3):
            c = a @ b.t()
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.1:   # the same warm-up as profiles/gemm_microbench.py
            for _ in range(5):
                c = a @ b.t()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            c = a @ b.t()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20
        tot_t += t
        tot_f += 2.0 * M * N * K
        print(f"hipBLASLt fp16 {name:5s} M={M} N={N} K={K}: {t*1e3:8.1f} us  {2.0*M*N*K/t/1e9:8.1f} TFLOP/s")
    print(f"layer total {tot_t*1e3:.1f} us -> {tot_f/tot_t/1e9:.1f} TFLOP/s")
