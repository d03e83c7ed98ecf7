import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from oracle import ref
from tests.conftest import load_pplhip
from tests.test_gpu_model import generate_both
m = load_pplhip()
for a8 in (0, 8):
    desc = ref.make_desc(hidden_dim=512, intermediate_dim=1024, num_layers=3, num_heads=8, num_kv_heads=8, vocab_size=2048,
                         max_position=512, cache_quant_bit=8, cache_quant_group=8, cache_layout=3, cache_mode=1, page_size=16,
                         weight_quant_bit=8, act_quant_bit=a8)
    rm = ref.RefModel(desc); rm.init_synthetic(79)
    ctx = m.Context(m.copy_desc(desc), max_running_batch=16, max_tokens_per_step=512)
    ctx.init_synthetic(0, 79)
    rm.kv_alloc(2048); ctx.kv_alloc(0, 2048)
    rng = np.random.RandomState(2)
    prompts = [rng.randint(3, 2048, size=n) for n in (40, 3, 129, 1, 16, 77)]
    res = generate_both(m, ctx, [rm], desc, prompts, 4, 2048)
    errs = [float(np.abs(r[0] - r[1]).max()) / max(1.0, float(np.abs(r[1]).max())) for r in res]
    print(f"tp 1 act_quant {a8}: errs {['%.2e' % e for e in errs]}", flush=True)
    ctx.close()
