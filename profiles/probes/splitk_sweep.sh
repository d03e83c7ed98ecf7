#!/bin/bash
# Round 4: how many split-K slabs should the half-height W8A16 kernels use at M = 16 .. 128?  (7B layer shapes, HBM-cold weights)
# usage (GPU box, repo root): bash profiles/probes/splitk_sweep.sh > gpurun_out/splitk_sweep.log
for sp in 1 2 3 4 6 8; do
  echo "== PPLHIP_GEMM_SPLITK=$sp (row-major kernels) / PPLHIP_FRAG_SPLITK=$sp (fragment-major kernel)"
  PPLHIP_FRAG_DBG=4 PPLHIP_GEMM_SPLITK=$sp PPLHIP_FRAG_SPLITK=$sp python profiles/frag_microbench.py 16 32 64 128 2>&1 | grep "M=" | sed 's/eq [0-9. ]*% //g'
done
