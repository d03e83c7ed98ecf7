"""debug: tiny TP group on one device, overlap schedule, repeated; prints where the logits leave the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import numpy as np
from oracle import ref
from tests.conftest import load_pplhip
from tests import test_gpu_tp as T
m = load_pplhip()
tp = int(sys.argv[1]) if len(sys.argv) > 1 else 2
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
os.environ["PPLHIP_TP_OVERLAP"] = "1"
os.environ["PPLHIP_TP_OVERLAP_MIN_TOKENS"] = "2"
desc = ref.make_desc(hidden_dim=512, intermediate_dim=1024, num_layers=3, num_heads=8, num_kv_heads=8, vocab_size=2048,
                     max_position=512, cache_quant_bit=8, cache_quant_group=8, cache_layout=3, cache_mode=mode,
                     page_size=16 if mode else 0, weight_quant_bit=8)
for rep in range(6):
    g = T.Group(m, desc, tp, max_batch=16, max_tokens=512, kv_tokens=2048)
    g.synthetic(31 + tp)
    rng = np.random.RandomState(tp)
    prompts = [rng.randint(3, 2048, size=n) for n in (40, 3, 129, 1, 16, 77)]
    try:
        res = T.generate(g, prompts, 4)
    except Exception as e:
        print("rep", rep, "EXC", repr(e)[:300]); g.close(); continue
    for s, (got, want, gtok) in enumerate(res):
        d = np.abs(got - want)
        bad_rows = np.nonzero(d.max(1) > 8e-3 * max(1, np.abs(want).max()))[0]
        print("rep", rep, "step", s, "maxerr", float(d.max()), "bad rows", bad_rows.tolist(), "finite", bool(np.isfinite(got).all()))
    g.close()
