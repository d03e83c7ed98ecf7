#!/bin/bash
# SQ / L2 counters of the M=1024 layer GEMMs (profiles/gemm_microbench.py); one rocprofv3 --pmc pass per counter group.
# usage (GPU box, from the repo root): bash profiles/gemm_pmc.sh <tag>   -> gpurun_out/gemm_pmc_<tag>_<group>.csv
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${1:-x}
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum FETCH_SIZE" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 600 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc_$i -- python $R/profiles/gemm_microbench.py 1024 > /tmp/pmc_$i.log 2>&1
  db=$(find /tmp/pmc_$i -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/profiles/summarize_rocpd.py pmc $db $R/gpurun_out/gemm_pmc_${tag}_$i.csv; else echo "no db for group $i"; tail -5 /tmp/pmc_$i.log; fi
done
