#!/bin/bash
# SQ counters of the M=1024 layer GEMMs for one kernel selection: bash profiles/gemm_pmc2.sh <tag> [env assignments...]
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${1:-x}; shift
for a in "$@"; do export "$a"; done
i=0
for grp in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 600 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc_$i -- python $R/profiles/gemm_microbench.py 1024 > /tmp/pmc_$i.log 2>&1
  db=$(find /tmp/pmc_$i -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/profiles/summarize_rocpd.py pmc $db $R/gpurun_out/gemm_pmc_${tag}_$i.csv; else echo "no db for group $i"; tail -5 /tmp/pmc_$i.log; fi
done
