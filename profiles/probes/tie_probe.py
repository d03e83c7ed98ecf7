import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from oracle import ref
from tests.conftest import load_pplhip
m = load_pplhip()
DIMS = dict(hidden_dim=4096, intermediate_dim=11008, num_layers=32, num_heads=32, num_kv_heads=32, vocab_size=32000)
desc = ref.make_desc(max_position=4096, cache_quant_bit=8, cache_quant_group=8, cache_layout=3, cache_mode=0, weight_quant_bit=8, weight_quant_group=128, **DIMS)
ctx = m.Context(m.copy_desc(desc), max_running_batch=8, max_tokens_per_step=4096)
ctx.init_synthetic(0, 1234)
assert m.lib().pplhip_rank_tie_output(ctx.h, 0, 7, 1234, float(sys.argv[1]) if len(sys.argv) > 1 else 8.0) == 0
ctx.kv_alloc(0, 8 * 1024)
rng = np.random.RandomState(0)
for L in (16, 512):
    n = 6
    toks = rng.randint(3, 32000, size=n * L).astype(np.int64)
    seq = np.arange(n + 1) * L
    ctx.set_inputs(0, m.make_step(toks, seq, np.zeros(n, dtype=np.int64), np.arange(n) * 1024, 0))
    ctx.run(0)
    lg = ctx.copy_logits(n)
    last = toks[seq[1:] - 1]
    srt = np.sort(lg, -1)
    print("L", L, "argmax==last+7:", (lg.argmax(-1) == (last + 7) % 32000), "margin/scale", (srt[:, -1] - srt[:, -2]) / np.abs(lg).max(), "scale", np.abs(lg).max(),
          "logit of last+7 rank", [(lg[i] > lg[i, (last[i] + 7) % 32000]).sum() for i in range(n)])
