import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from oracle import ref
from tests.conftest import load_pplhip
from tests.test_gpu_tp import Group, generate
m = load_pplhip()
for tp in (1, 2, 4):
  for a8 in (0, 8):
    desc = ref.make_desc(hidden_dim=512, intermediate_dim=1024, num_layers=3, num_heads=8, num_kv_heads=8, vocab_size=2048,
                         max_position=512, cache_quant_bit=8, cache_quant_group=8, cache_layout=3, cache_mode=1, page_size=16,
                         weight_quant_bit=8, act_quant_bit=a8)
    if tp == 1:
        continue
    g = Group(m, desc, tp, max_batch=16, max_tokens=512, kv_tokens=2048)
    g.synthetic(77 + tp)
    rng = np.random.RandomState(tp)
    prompts = [rng.randint(3, 2048, size=n) for n in (40, 3, 129, 1, 16, 77)]
    res = generate(g, prompts, 4)
    errs = [float(np.abs(got - want).max()) / max(1.0, float(np.abs(want).max())) for got, want, _ in res]
    agree = [float((gtok == want.argmax(-1)).mean()) for got, want, gtok in res]
    print(f"tp {tp} act_quant {a8}: errs {['%.2e' % e for e in errs]} greedy agreement {agree}", flush=True)
    g.close()
