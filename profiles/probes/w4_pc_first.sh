#!/bin/bash
# round 5: first contact of k_gemm_pc.hip -- parity subset, then A/B of the 70B/TP8 layer GEMMs at M = 256 (HBM-cold microbench)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "test_linear and not gemv and not w8" 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_config34_shape.py -x -q -m gpu -k "config4_w4a16" 2>&1 | tail -5
for pc in 0 1 0 1; do
  echo "== PPLHIP_GEMM_PC=$pc"
  PPLHIP_GEMM_PC=$pc timeout 300 python profiles/gemm_microbench.py 256 4 70b-tp8 2>&1 | grep -v amdgpu.ids
done
