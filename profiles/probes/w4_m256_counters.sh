#!/bin/bash
# Round 4: what bounds the W4A16 tile kernel at M = 256 (config 4)?  SQ / TCC counters of w13 (N 7168, K 8192) and w2 (N 8192, K 3584) of the
# 70B / TP8 slice, one rocprofv3 --pmc pass per counter group.   usage (GPU box, repo root): bash profiles/probes/w4_m256_counters.sh
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for shape in w13 w2; do
  i=0
  for grp in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY"; do
    i=$((i+1)); rm -rf /tmp/wc_$i
    timeout 300 rocprofv3 --pmc $grp --kernel-trace -d /tmp/wc_$i -- python $R/profiles/gemm_microbench.py 256 4 70b-tp8 $shape > /tmp/wc_$i.log 2>&1
    db=$(find /tmp/wc_$i -name "*.db" | head -1)
    if [ -n "$db" ]; then python $R/profiles/summarize_rocpd.py pmc $db $R/gpurun_out/r04_w4_m256_counters_${shape}_$i.csv; else echo "$shape group $i ($grp): no output" >> $R/gpurun_out/w4_m256_counters.err; tail -3 /tmp/wc_$i.log >> $R/gpurun_out/w4_m256_counters.err; fi
  done
done
cat $R/gpurun_out/r04_w4_m256_counters_*.csv | grep -v "^kernel" | grep gemm_dma | cut -d, -f2- | sed 's/^[^"]*"//' 
