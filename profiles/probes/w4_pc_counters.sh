#!/bin/bash
# Round 5: SQ / TCC counters of gemm_w4_pc_kernel on the 70B / TP8 shapes at M = 256, one rocprofv3 --pmc pass per counter group
# (profiles/probes/w4_m256_counters.sh of round 4 for the new kernel).  usage (GPU box, repo root): bash profiles/probes/w4_pc_counters.sh [tag]
R=$GRAFT_REPO_ROOT
TAG=${1:-r05_w4_m256_counters}
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
for shape in ${SHAPES:-w13 w2}; do
  i=0
  for grp in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY"; do
    i=$((i+1)); rm -rf /tmp/wc_$i
    timeout 300 rocprofv3 --pmc $grp --kernel-trace -d /tmp/wc_$i -- python $R/profiles/gemm_microbench.py ${MROWS:-256} 4 70b-tp8 $shape > /tmp/wc_$i.log 2>&1
    db=$(find /tmp/wc_$i -name "*.db" | head -1)
    if [ -n "$db" ]; then python $R/profiles/summarize_rocpd.py pmc $db $R/gpurun_out/$TAG/${shape}_$i.csv; else echo "$shape group $i ($grp): no output" >> $R/gpurun_out/$TAG/errors.txt; tail -3 /tmp/wc_$i.log >> $R/gpurun_out/$TAG/errors.txt; fi
  done
done
cat $R/gpurun_out/$TAG/*.csv | grep -v "^kernel" | grep "gemm_w4_pc\|gemm_dma" | sed 's/^"[^"]*",//' 
