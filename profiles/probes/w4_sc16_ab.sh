#!/bin/bash
# Round 4 (last session): W4A16 tile kernel with a 4-KiB scale area (PPLHIP_GEMM_W4_SC16=1: three blocks per CU) vs 16 KiB (two blocks per CU)
R=${GRAFT_REPO_ROOT:-/root/repo}
for M in 256 128; do
 for shape in w13 wqkv; do
  for v in "PPLHIP_GEMM_W4_SC16=0" "PPLHIP_GEMM_W4_SC16=1" "PPLHIP_GEMM_W4_SC16=0 PPLHIP_GEMM_SPLITK=6" "PPLHIP_GEMM_W4_SC16=1 PPLHIP_GEMM_SPLITK=6" "PPLHIP_GEMM_W4_SC16=1 PPLHIP_GEMM_SPLITK=8" "PPLHIP_GEMM_W4_SC16=1 PPLHIP_GEMM_SPLITK=5"; do
    echo "== M=$M $v"; env $v python $R/profiles/gemm_microbench.py $M 4 70b-tp8 $shape 2>&1 | grep "M="
  done
 done
done
L2="--no-cpu-baseline --no-serving-leg --no-i8i8-leg --prefill-sample 0 --ragged-steps 0 --breakdown-steps 0"
for v in 0 1 0 1; do
  echo "== config 4 step, PPLHIP_GEMM_W4_SC16=$v"
  PPLHIP_GEMM_W4_SC16=$v python $R/bench.py --model llama2-70b --weight-quant 4 --batch 256 --kv-len 2048 --emulate-tp 8 $L2 2>/dev/null | python -c 'import sys,json; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("ms_per_step", r["ms_per_step"])'
done
