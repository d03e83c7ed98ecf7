#!/bin/bash
# Round 4 (last session): four-stage ring on split-K launches of at most one block per CU (PPLHIP_GEMM_SPLIT_DEEP=1, the default) vs two stages (0)
R=${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "256 8 7b" "384 8 7b" "512 8 13b-tp2" "256 4 70b-tp8" "64 4 70b-tp8" "64 0 7b" "256 0 70b-tp8" "1024 8 7b-tp8"; do
  for d in 0 1; do echo "== $cfg PPLHIP_GEMM_SPLIT_DEEP=$d"; PPLHIP_GEMM_SPLIT_DEEP=$d python $R/profiles/gemm_microbench.py $cfg 2>&1 | grep -E "M=|layer"; done
done
L2="--no-cpu-baseline --no-serving-leg --no-i8i8-leg --prefill-sample 0 --ragged-steps 0 --breakdown-steps 0"
fmt='import sys,json; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("ms_per_step", r["ms_per_step"])'
for d in 0 1 0 1; do
  echo "== config 3 per rank (13B W8A16 / TP2, 512 rows), PPLHIP_GEMM_SPLIT_DEEP=$d"
  PPLHIP_GEMM_SPLIT_DEEP=$d python $R/bench.py --model llama2-13b --batch 512 --kv-len 1024 --emulate-tp 2 $L2 2>/dev/null | python -c "$fmt"
  echo "== config 4 per rank (70B W4A16 / TP8, 256 rows), PPLHIP_GEMM_SPLIT_DEEP=$d"
  PPLHIP_GEMM_SPLIT_DEEP=$d python $R/bench.py --model llama2-70b --weight-quant 4 --batch 256 --kv-len 2048 --emulate-tp 8 $L2 2>/dev/null | python -c "$fmt"
  echo "== 7B / TP8 slice, 1024 rows, PPLHIP_GEMM_SPLIT_DEEP=$d"
  PPLHIP_GEMM_SPLIT_DEEP=$d python $R/bench.py --emulate-tp 8 $L2 2>/dev/null | python -c "$fmt"
done
for d in 0 1 0 1; do PPLHIP_GEMM_SPLIT_DEEP=$d python $R/profiles/small_batch_latency.py 160 192 256 384 512 2>&1 | grep batch | sed "s/^/split_deep=$d /"; done
