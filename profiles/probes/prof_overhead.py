import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tests.conftest import load_pplhip
P = load_pplhip()
MK = dict(hidden_dim=4096, intermediate_dim=11008, num_layers=32, num_heads=32, num_kv_heads=32, vocab_size=32000)
KV, STEPS, WARM, B = 512, 16, 3, 1024
for prof in (0, 2, 1, 0, 2):
    desc = P.make_desc(max_position=2048, cache_quant_bit=8, cache_quant_group=8, cache_layout=3, cache_mode=0, weight_quant_bit=8, **MK)
    ctx = P.Context(desc, max_running_batch=B, max_tokens_per_step=8192, profiling=prof)
    ctx.init_synthetic(0, 1234)
    tl = KV + STEPS + WARM + 2
    ctx.kv_alloc(0, B * tl); ctx.kv_fill_synthetic(0, 9)
    tok = np.random.RandomState(0).randint(3, 32000, size=B).astype(np.int64)
    ci = np.arange(B, dtype=np.int64) * tl
    seq = np.arange(B + 1)
    def step(i, tok):
        ctx.set_inputs(0, P.make_step(tok, seq, np.full(B, KV + i), ci, B, req_list_changed=int(i == 0)))
        ctx.run(0)
        return ctx.sample(B, top_k=1, req_list_changed=(i == 0))[0].astype(np.int64)
    for i in range(WARM): tok = step(i, tok)
    ctx.sync(0); t0 = time.perf_counter()
    for i in range(WARM, WARM + STEPS): tok = step(i, tok)
    ctx.sync(0); dt = (time.perf_counter() - t0) / STEPS
    print(f"profiling {prof}: {dt*1e3:.3f} ms/step = {B/dt:.1f} tokens/s", flush=True)
    ctx.close()
