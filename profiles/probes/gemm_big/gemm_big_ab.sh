#!/bin/bash
# W8A16 layer GEMMs at M = 8192 / 4096: the four-wave 32 x 32 x 16 kernel (k_gemm_big.hip) against the eight-wave 256 x 256 kernel, same box, then parity
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for b in 0 8 4; do
  echo "== PPLHIP_GEMM_BIG=$b (rep $rep)"
  PPLHIP_GEMM_BIG=$b python profiles/gemm_microbench.py 8192 8 7b 2>&1 | grep -v amdgpu
  PPLHIP_GEMM_BIG=$b python profiles/gemm_microbench.py 4096 8 7b 2>&1 | grep "layer total"
done; done
python -m pytest tests/test_gpu_ops.py -x -q -k "linear" 2>&1 | tail -2
python -m pytest tests/test_gpu_config2_shape.py -x -q 2>&1 | tail -2
