#!/bin/bash
# Round 4: W4A16 128 (m) x 256 (n) tile kernel, 64 x 128 wave tiles (PPLHIP_GEMM_W4_BN256) against the 128 x 128 tiles (32 x 128 wave tiles)
# on the 70B / TP8 shapes, and config 4's per-rank step
R=${GRAFT_REPO_ROOT:-/root/repo}
for M in 256 128 512; do
  for v in "PPLHIP_GEMM_W4_BN256=0" "PPLHIP_GEMM_W4_BN256=1" "PPLHIP_GEMM_W4_BN256=1 PPLHIP_GEMM_W4_BN256_STAGES=3" "PPLHIP_GEMM_W4_BN256=1 PPLHIP_GEMM_W4_BN256_BLOCKS=512" "PPLHIP_GEMM_W4_BN256=2 PPLHIP_GEMM_W4_BN256_BLOCKS=512" "PPLHIP_GEMM_W4_BN256=2 PPLHIP_GEMM_W4_BN256_BLOCKS=128"; do
    echo "== M=$M $v"; env $v python $R/profiles/gemm_microbench.py $M 4 70b-tp8 2>&1 | grep "M="
  done
done
L2="--no-cpu-baseline --no-serving-leg --no-i8i8-leg --prefill-sample 0 --ragged-steps 0 --breakdown-steps 0"
for v in 0 1 0 1; do
  echo "== config 4 step, PPLHIP_GEMM_W4_BN256=$v"
  PPLHIP_GEMM_W4_BN256=$v python $R/bench.py --model llama2-70b --weight-quant 4 --batch 256 --kv-len 2048 --emulate-tp 8 $L2 2>/dev/null | python -c 'import sys,json; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("ms_per_step", r["ms_per_step"])'
done
