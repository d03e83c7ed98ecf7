#!/bin/bash
# Round 4: SQ / TCC counters of the M = 64 layer linears (profiles/gemv_microbench.py 8 64), one pass per counter group
# usage (GPU box, repo root): bash profiles/probes/half128_counters.sh
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $R/gpurun_out/rocprof_counters.txt 2>&1
i=0
for grp in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM"; do
  i=$((i+1)); rm -rf /tmp/hc_$i
  timeout 600 rocprofv3 --pmc $grp --kernel-trace -d /tmp/hc_$i -- python $R/profiles/gemv_microbench.py 8 64 > /tmp/hc_$i.log 2>&1
  db=$(find /tmp/hc_$i -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/profiles/summarize_rocpd.py pmc $db $R/gpurun_out/r04_half128_counters_$i.csv; else echo "group $i ($grp): no output" >> $R/gpurun_out/half128_counters.err; tail -3 /tmp/hc_$i.log >> $R/gpurun_out/half128_counters.err; fi
done
