#!/bin/bash
# Round 4 (last session): W4A16 at M = 256 with the twelve-wave block (eight consumers: two groups, one per k-step of a tile; PPLHIP_GEMM_WL=6) and FEWER K slabs
R=${GRAFT_REPO_ROOT:-/root/repo}
for shape in w13 w2; do
  for v in "X=0" "PPLHIP_GEMM_SPLITK=2" "PPLHIP_GEMM_SPLITK=2 PPLHIP_GEMM_WL=6" "PPLHIP_GEMM_SPLITK=2 PPLHIP_GEMM_WL=6 PPLHIP_GEMM_STAGES=3" "PPLHIP_GEMM_SPLITK=2 PPLHIP_GEMM_WL=5" "PPLHIP_GEMM_SPLITK=1 PPLHIP_GEMM_WL=6" "PPLHIP_GEMM_SPLITK=3 PPLHIP_GEMM_WL=6" "PPLHIP_GEMM_SPLITK=4 PPLHIP_GEMM_WL=6"; do
    echo "== M=256 $shape $v"; env $v python $R/profiles/gemm_microbench.py 256 4 70b-tp8 $shape 2>&1 | grep "M="
  done
done
