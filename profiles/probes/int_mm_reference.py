import torch
SHAPES = [("wqkv", 12288, 4096), ("wo", 4096, 4096), ("w13", 22016, 4096), ("w2", 4096, 11008)]
for M in (1024, 8192):
    tt=tf=0
    for name, N, K in SHAPES:
        a = torch.randint(-127, 128, (M, K), device="cuda", dtype=torch.int8)
        b = torch.randint(-127, 128, (N, K), device="cuda", dtype=torch.int8)
        bt = b.t()
        try:
            for _ in range(3): c = torch._int_mm(a, bt)
        except Exception as e:
            print("int_mm failed:", repr(e)[:200]); raise SystemExit
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): c = torch._int_mm(a, bt)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20
        tt+=t; tf+=2.0*M*N*K
        print(f"hipBLASLt int8 {name:5s} M={M} N={N} K={K}: {t*1e3:8.1f} us  {2.0*M*N*K/t/1e9:8.1f} TOP/s")
    print(f"layer total {tt*1e3:.1f} us -> {tf/tt/1e9:.1f} TOP/s")
