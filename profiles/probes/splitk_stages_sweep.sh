#!/bin/bash
# Round 4 (last session): ring depth of the tile kernels WITH K slabs.  launch_linear forced two stages whenever splits > 1 and the
# PPLHIP_GEMM_STAGES override was applied BEFORE that line, so the stage sweeps of r04_w4_m256_sweep.log never ran 3 / 4 stages on the
# split shapes (their three rows are identical).  With the override honoured:
R=${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "256 4 70b-tp8" "128 4 70b-tp8" "512 4 70b-tp8" "256 8 7b" "128 8 7b-tp8" "512 8 13b-tp2" "256 0 70b-tp8"; do
  for st in 0 3 4; do
    echo "== $cfg PPLHIP_GEMM_STAGES=$st (0 = default)"
    if [ $st = 0 ]; then python $R/profiles/gemm_microbench.py $cfg 2>&1 | grep -E "M=|layer"; else PPLHIP_GEMM_STAGES=$st python $R/profiles/gemm_microbench.py $cfg 2>&1 | grep -E "M=|layer"; fi
  done
done
