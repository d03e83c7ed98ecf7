#!/usr/bin/env python3
"""One-off robustness run (round 5): pplhip_op_linear / pplhip_op_linear_swiglu on random shapes drawn around the dispatcher's thresholds
(k_gemm.hip launch_linear: GEMV / half-height tiles / 128 x 128 ring with and without split-K / 128 x 64 W4 tiles / 128 x 384 /
256 x 256; ragged N, K multiples of the smallest legal unit), every case against the oracle with the tolerances of tests/test_gpu_ops.py.
usage: python profiles/probes/linear_fuzz.py [seconds] [seed]"""
import os, sys, time, traceback
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tests.test_gpu_ops as T

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
MS = [1, 2, 3, 4, 5, 8, 15, 16, 17, 31, 32, 33, 63, 64, 65, 80, 96, 127, 128, 129, 200, 255, 256, 257, 300, 511, 512, 513, 700, 1023, 1024, 1025, 1040, 1300, 2048, 2100]
t0, n, bad = time.time(), 0, []
while time.time() - t0 < budget:
    wq = int(rng.choice([0, 8, 4]))
    M = int(rng.choice(MS)) if rng.rand() < 0.8 else int(rng.randint(1, 2200))
    kunit = 128 if (wq == 4 and rng.rand() < 0.8) else (32 if wq == 4 else int(rng.choice([16, 64, 128]) if wq == 8 else rng.choice([8, 64, 128])))
    K = kunit * int(rng.randint(1, max(2, 9000 // kunit)))
    if rng.rand() < 0.5: K = kunit * int(rng.randint(1, max(2, 1500 // kunit)))
    swiglu = rng.rand() < 0.3
    if swiglu and wq == 4 and K % 128: K = 128 * max(1, K // 128)   # (the fused-SwiGLU test quantises in groups of 128)
    if swiglu:
        inter = 4 * int(rng.randint(1, 3000))
        if M * inter * K > 3e10: continue
        case = ("swiglu", wq, M, inter, K)
    else:
        N = 4 * int(rng.randint(2, 3200))
        if rng.rand() < 0.3: N = 128 * int(rng.randint(1, 100))
        if M * N * K > 3e10: continue
        case = ("linear", wq, M, N, K)
    try:
        if swiglu: T.test_linear_swiglu_fused(wq, M, case[3], K)
        else: T.test_linear(wq, M, case[3], K)
    except AssertionError as e:
        bad.append((case, str(e)[:300]))
        print("FAIL", case, str(e)[:300], flush=True)
    except Exception as e:
        bad.append((case, repr(e)[:300]))
        print("ERROR", case, repr(e)[:300], flush=True)
    n += 1
print(f"{n} cases in {time.time() - t0:.0f} s, {len(bad)} failures")
for b in bad: print(b)
