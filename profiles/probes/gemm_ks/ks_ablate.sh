#!/bin/bash
# k-split W8A16 kernel (csrc/k_gemm_ks.hip): what bounds its K loop?  Diagnosis builds (WRONG results) of the library with parts of the loop
# compiled out, timed on the 7B / TP8 slice shapes (w13: 128 x 96 tiles, wqkv: 64 x 96).  usage (GPU box, repo root): bash profiles/probes/ks_ablate.sh
cd $GRAFT_REPO_ROOT/ppl.llm.serving_amd/csrc
for a in 0 1 2 4 8 6 3; do
  make -s -j16 EXTRA=-DKS_ABL=$a > /dev/null 2>&1
  echo "== KS_ABL=$a (1 no MFMAs, 2 no ring refills, 4 no fragment re-reads, 8 no loop barriers)"
  python $GRAFT_REPO_ROOT/profiles/gemm_microbench.py 1024 8 7b-tp8 wqkv 2>&1 | grep "M="
  python $GRAFT_REPO_ROOT/profiles/gemm_microbench.py 1024 8 7b-tp8 w13 2>&1 | grep "M="
done
make -s -j16 > /dev/null 2>&1
