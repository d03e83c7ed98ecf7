import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.conftest import load_pplhip
from oracle import ref
m = load_pplhip()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 2
desc = ref.make_desc(hidden_dim=4096, intermediate_dim=11008, num_layers=L, num_heads=32, num_kv_heads=32, vocab_size=32000,
                     max_position=256, cache_quant_bit=8, cache_quant_group=8, cache_layout=3, cache_mode=0, weight_quant_bit=8)
B, ctx_len = 1024, 40
rng = np.random.RandomState(1)
tok = rng.randint(3, 32000, size=B).astype(np.int64)
start_pos = rng.randint(1, ctx_len, size=B).astype(np.int64)
cache_idx = (np.arange(B) * (ctx_len + 1)).astype(np.int64)
out = []
for fused in (0, 1, 1):
    os.environ["PPLHIP_FUSED_DECODE"] = str(fused)
    ctx = m.Context(m.copy_desc(desc), max_running_batch=B, max_tokens_per_step=B, profiling=True)
    ctx.init_synthetic(0, 9); ctx.kv_alloc(0, B * (ctx_len + 1)); ctx.kv_fill_synthetic(0, 3)
    ctx.set_inputs(0, m.make_step(tok, np.arange(B + 1), start_pos, cache_idx, B, 0, req_list_changed=1))
    ctx.run(0)
    out.append(ctx.copy_logits(B).copy())
    ctx.close()
d = np.abs(out[0] - out[1]).max(axis=1)
print("plain vs fused: max", d.max(), "rows differing", (d > 0).sum(), "first half", (d[:512] > 0).sum(), "second", (d[512:] > 0).sum())
print("logit scale", np.abs(out[0]).max())
print("fused vs fused:", np.abs(out[1] - out[2]).max())
print("rows differing idx sample", np.nonzero(d > 0)[0][:20])
