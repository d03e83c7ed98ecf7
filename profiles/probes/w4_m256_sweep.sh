#!/bin/bash
# Round 4: split-K slabs / stages of the W4A16 tile kernel on the 70B / TP8 layer shapes at M = 256 (config 4) and M = 64
# usage (GPU box, repo root): bash profiles/probes/w4_m256_sweep.sh
for M in 256 64; do
  echo "== M=$M default"; python profiles/gemm_microbench.py $M 4 70b-tp8 2>&1 | grep "M="
  for sp in 1 2 3 4 6 8; do echo "== M=$M PPLHIP_GEMM_SPLITK=$sp"; PPLHIP_GEMM_SPLITK=$sp python profiles/gemm_microbench.py $M 4 70b-tp8 2>&1 | grep "M="; done
  for st in 2 3 4; do echo "== M=$M PPLHIP_GEMM_STAGES=$st"; PPLHIP_GEMM_STAGES=$st python profiles/gemm_microbench.py $M 4 70b-tp8 2>&1 | grep "M="; done
done
