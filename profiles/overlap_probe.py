#!/usr/bin/env python3
"""Probe: do the HBM-bound decode attention and the MFMA-bound GEMMs of two independent half-batches overlap when they
run on two HIP streams?  Two contexts (batch 512 each) driven from two threads vs one context (batch 1024)."""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.conftest import load_pplhip
P = load_pplhip()
MK = dict(hidden_dim=4096, intermediate_dim=11008, num_layers=32, num_heads=32, num_kv_heads=32, vocab_size=32000)
KV, STEPS = 512, 8

def make(B):
    desc = P.make_desc(max_position=2048, cache_quant_bit=8, cache_quant_group=8, cache_layout=3, cache_mode=0, weight_quant_bit=8, **MK)
    ctx = P.Context(desc, max_running_batch=B, max_tokens_per_step=max(B, 1024))
    ctx.init_synthetic(0, 1234)
    ctx.kv_alloc(0, B * (KV + STEPS + 4)); ctx.kv_fill_synthetic(0, 9)
    return ctx

def run(ctx, B, steps, stagger=0.0):
    time.sleep(stagger)
    tok = np.random.RandomState(0).randint(3, 32000, size=B).astype(np.int64)
    ci = np.arange(B, dtype=np.int64) * (KV + STEPS + 4)
    for i in range(steps):
        st = P.make_step(tok, np.arange(B + 1), np.full(B, KV + i), ci, B, req_list_changed=int(i == 0))
        ctx.set_inputs(0, st); ctx.run(0)
        tok = ctx.sample(B, top_k=1, req_list_changed=(i == 0))[0].astype(np.int64)

one = make(1024)
run(one, 1024, 2)
t0 = time.perf_counter(); run(one, 1024, STEPS); t1 = time.perf_counter()
print(f"one context  B=1024: {1024 * STEPS / (t1 - t0):9.1f} tok/s  ({(t1 - t0) / STEPS * 1e3:.2f} ms/step)")
one.close()
a, b = make(512), make(512)
run(a, 512, 2); run(b, 512, 2)
t0 = time.perf_counter(); run(a, 512, STEPS); t1 = time.perf_counter()
print(f"one context  B=512 : {512 * STEPS / (t1 - t0):9.1f} tok/s  ({(t1 - t0) / STEPS * 1e3:.2f} ms/step)")
for stagger in (0.0, 0.006):
    ta = threading.Thread(target=run, args=(a, 512, STEPS)); tb = threading.Thread(target=run, args=(b, 512, STEPS, stagger))
    t0 = time.perf_counter(); ta.start(); tb.start(); ta.join(); tb.join(); t1 = time.perf_counter()
    print(f"two contexts B=512x2 (stagger {stagger*1e3:.0f} ms): {1024 * STEPS / (t1 - t0 - stagger):9.1f} tok/s  ({(t1 - t0) / STEPS * 1e3:.2f} ms per double-step)")
