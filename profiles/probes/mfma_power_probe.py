#!/usr/bin/env python3
"""pure MFMA streams, fp16, 16x16x32 vs 32x32x16, on random / integer-valued / zero operands, 1 / 2 / 3 waves per SIMD (round 4)"""
import ctypes as C, os, torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmfma_power_probe.so"))
lib.mfma_probe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
n = 1 << 19
data = {"random N(0,1) x N(0,1)": ((torch.randn(n * 2, device="cuda")).half(), (torch.randn(n * 2, device="cuda")).half()),
        "int8-valued x N(0,0.5)": (torch.randint(-127, 128, (n * 2,), device="cuda").half(), (torch.randn(n * 2, device="cuda") * 0.5).half()),
        "zeros": (torch.zeros(n * 2, device="cuda").half(), torch.zeros(n * 2, device="cuda").half())}
out = torch.zeros(1 << 20, device="cuda")
for wps in (1, 2, 3):
    blocks = 256 * wps
    for name, (a, b) in data.items():
        for shape in (16, 32):
            per_it = (16 * 16 * 32 * 2 * 16) if shape == 16 else (32 * 32 * 16 * 2 * 8)   # flop per wave and iteration
            iters = 20000 if shape == 16 else 20000
            call = lambda: lib.mfma_probe(shape, blocks, iters, a.data_ptr(), b.data_ptr(), out.data_ptr(), None)
            call(); torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); call(); call(); e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 2)
            fl = per_it * iters * blocks * 4
            print(f"waves/SIMD {wps}  {name:24s} mfma {shape}x{shape}: {best:8.2f} ms  {fl / best / 1e9:8.1f} TFLOP/s", flush=True)
