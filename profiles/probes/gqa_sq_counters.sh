#!/bin/bash
# grouped-query decode attention at config 4's shape: where do the cycles go?  SQ counters in separate passes (--kernel-trace only).
# usage (GPU box, repo root): bash profiles/probes/gqa_sq_counters.sh <out-prefix>
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-gqa_sq}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/gqa_one.py <<'PY'
import sys, os
sys.argv = ["x", "/dev/null"]
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "profiles"))
import roofline_sweep as rs
print(rs.run(256, 2048, 8, 1, 1, iters=16, warm=4))
PY
i=0
for c in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
         "SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH" \
         "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TD_BUSY_avr" "SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_MUL_F16 SQ_INSTS_VALU_FMA_F16"; do
  i=$((i+1)); rm -rf /tmp/prof_q
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d /tmp/prof_q -- python /tmp/gqa_one.py > /tmp/prof_q.log 2>&1
  db=$(find /tmp/prof_q -name "*.db" | head -1)
  [ -n "$db" ] && python $R/profiles/summarize_rocpd.py pmc $db ${OUT}_$i.csv || { echo "pass $i ($c) failed:"; tail -3 /tmp/prof_q.log; }
done
grep -h attn_decode_gqa ${OUT}_*.csv | awk -F'",' '{print $2}' | cut -d, -f1-3,6
