#!/bin/bash
# NOTE: needs profiles/probes/decode_order_longest_first.patch applied and built (measured slower, not adopted: profiles/r05_decode_order_ab.log)
# round 5: decode attention dispatched longest context first (pplhip_set_inputs' decode_order) against batch order (PPLHIP_DECODE_ORDER=0):
# the ragged-batch leg of bench.py (samples_1024-shaped context lengths 4 .. 1024 in one step) and the uniform headline step, A/B/A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; O=gpurun_out/decode_order_ab.log; : > $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "attention_decode or attention_mixed" 2>&1 | tail -2 >> $O
for rep in 1 2; do for o in 0 1; do
  PPLHIP_DECODE_ORDER=$o python bench.py --steps 8 --warmup 3 --no-serving-leg --no-i8i8-leg --ragged-steps 8 > /tmp/b.json 2>/tmp/b.err
  python3 - $o >> $O <<'PY'
import json, sys
r = json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
rb = r.get("ragged_batch", {})
print(f"PPLHIP_DECODE_ORDER={sys.argv[1]}: uniform {r['ms_per_step']} ms/step, attention frac {r['roofline']['frac']}; ragged {rb.get('ms_per_step')} ms/step, {rb.get('tokens_per_s')} tokens/s, attention frac {rb.get('attn_decode_frac_of_8TBps')}")
PY
done; done
cat $O
