#!/bin/bash
# Round 4 (last session): RMSNorm on 512 / 1024 threads per row for steps of 5..512 rows (PPLHIP_RMSNORM_WIDE_MAX_ROWS=512, default) vs 256 threads (=0)
R=${GRAFT_REPO_ROOT:-/root/repo}
L2="--no-cpu-baseline --no-serving-leg --no-i8i8-leg --prefill-sample 0 --ragged-steps 0 --breakdown-steps 0"
fmt='import sys,json; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("ms_per_step", r["ms_per_step"])'
for v in 0 512 0 512; do
  echo "== config 4 per rank, PPLHIP_RMSNORM_WIDE_MAX_ROWS=$v"
  PPLHIP_RMSNORM_WIDE_MAX_ROWS=$v python $R/bench.py --model llama2-70b --weight-quant 4 --batch 256 --kv-len 2048 --emulate-tp 8 $L2 2>/dev/null | python -c "$fmt"
done
for v in 0 512 0 512; do PPLHIP_RMSNORM_WIDE_MAX_ROWS=$v python $R/profiles/small_batch_latency.py 8 32 64 128 256 512 2>&1 | grep batch | sed "s/^/wide_max_rows=$v /"; done
