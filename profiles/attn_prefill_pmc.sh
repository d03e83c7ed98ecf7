set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16"; do
  i=$((i+1)); rm -rf /tmp/pa_$i
  timeout 600 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pa_$i -- python $R/profiles/attn_prefill_microbench.py 1 8192 0 0 > /tmp/pa_$i.log 2>&1
  db=$(find /tmp/pa_$i -name "*.db" | head -1)
  [ -n "$db" ] && python $R/profiles/summarize_rocpd.py pmc $db $R/gpurun_out/attn_prefill_pmc_$i.csv || tail -3 /tmp/pa_$i.log
done
