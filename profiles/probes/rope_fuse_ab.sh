#!/bin/bash
# NOTE: needs profiles/probes/rope_folded_into_decode_attention.patch applied and built (bit-identical, measured slower, not adopted: profiles/r05_rope_fuse_ab.log)
# round 5: RoPE + KV write folded into the decode attention kernel (default) against the two-kernel path (PPLHIP_ROPE_FUSE=0):
# parity first (operator + model level, bit for bit), then the decode step at batch 1 .. 128 and the headline step, A/B/A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; O=gpurun_out/rope_fuse_ab.log; : > $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q -k "folded or rope or attention_decode or attention_mixed" 2>&1 | tail -4 >> $O
for rep in 1 2; do for f in 0 1; do
  echo "== PPLHIP_ROPE_FUSE=$f" >> $O
  PPLHIP_ROPE_FUSE=$f python profiles/small_batch_latency.py 1 2 4 8 16 32 64 128 2>&1 | grep "^batch" >> $O
done; done
for f in 0 1 0 1; do
  PPLHIP_ROPE_FUSE=$f python bench.py --steps 12 --warmup 4 --no-serving-leg --no-i8i8-leg --no-cpu-baseline --ragged-steps 4 > /tmp/b.json 2>/tmp/b.err
  python3 - $f >> $O <<'PY'
import json, sys
r = json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
rb = r.get("ragged_batch", {})
print(f"PPLHIP_ROPE_FUSE={sys.argv[1]}: headline {r['ms_per_step']} ms/step ({r['value']} tokens/s), attention frac {r['roofline']['frac']}; ragged {rb.get('ms_per_step')} ms/step")
PY
done
cat $O
