"""Which of two batch orders of the same prompts disagrees with running each prompt alone?  (diagnosis of a failing
bit-for-bit permutation test)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.conftest import load_pplhip
from tests import test_gpu_properties as T
m = load_pplhip()
rng = np.random.RandomState(5)
prompts = [rng.randint(3, 32000, size=k).astype(np.int64) for k in (37, 1, 130, 64, 5, 17)]
L = int(os.environ.get("LAYERS", "32"))
ctx = T.make_ctx(m, num_layers=L)
a0, a1 = T.run_two_steps(m, ctx, prompts, 0, 16, 8192)
order = np.array([3, 0, 5, 1, 4, 2])
c0, c1 = T.run_two_steps(m, ctx, prompts, 0, 16, 8192, order=order)
inv = np.argsort(order)
c0, c1 = c0[inv], c1[inv]
print("layers", L, "prefill a vs c per prompt:", np.abs(c0 - a0).max(1))
for i, p in enumerate(prompts):
    s0, s1 = T.run_two_steps(m, ctx, [p], 0, 16, 8192)
    print(f"prompt {i} len {len(p)}: alone vs a {np.abs(s0[0]-a0[i]).max():.4f} / decode {np.abs(s1[0]-a1[i]).max():.4f};"
          f" alone vs c {np.abs(s0[0]-c0[i]).max():.4f} / decode {np.abs(s1[0]-c1[i]).max():.4f}")
