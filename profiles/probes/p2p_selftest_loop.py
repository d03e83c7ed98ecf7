"""debug: repeated context creation (= repeated self-tests of the direct collectives) on one device."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
os.environ["PPLHIP_VERBOSE"] = "1"
from tests.conftest import load_pplhip
m = load_pplhip()
tp = int(sys.argv[1]); reps = int(sys.argv[2])
desc = m.make_desc(hidden_dim=512, intermediate_dim=1024, num_layers=1, num_heads=8, num_kv_heads=8, vocab_size=2048, max_position=512,
                   cache_quant_bit=8, cache_quant_group=8, weight_quant_bit=8)
bad = 0
for i in range(reps):
    try:
        ctx = m.Context(desc, max_running_batch=16, max_tokens_per_step=512, n_local_ranks=tp, device_ids=[0] * tp)
        ctx.close()
    except Exception as e:
        bad += 1
print("tp", tp, "failed inits", bad, "of", reps, "xalloc", os.environ.get("PPLHIP_P2P_XALLOC", "1"))
