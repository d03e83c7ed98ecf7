#!/bin/bash
# W8A16 layer GEMMs at M = 8192 (gemm_w8_dma256_kernel) and the vendor library on the same shapes: HBM traffic and L2 hit rates per launch
# against the algorithmic bytes (x M K 2 + w N K + y M N 2: wqkv 67 + 50 + 201 MB).  usage (GPU box, repo root): bash profiles/probes/gemm_bigm_traffic.sh
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/gemm_bigm
cd /tmp && export TMPDIR=/tmp
cat > /tmp/vend.py <<'PY'
import torch
for name, N, K in (("wqkv", 12288, 4096), ("w2", 4096, 11008)):
    a = torch.randn(8192, K, device="cuda", dtype=torch.float16); b = torch.randn(N, K, device="cuda", dtype=torch.float16)
    for _ in range(12): c = a @ b.t()
    torch.cuda.synchronize()
PY
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU"; do
  tag=$(echo $c | tr ' ' '_')
  for which in ours vendor; do
    rm -rf /tmp/prof_b
    if [ $which = ours ]; then cmd="python $R/profiles/gemm_microbench.py 8192 8 7b"; else cmd="python /tmp/vend.py"; fi
    timeout 600 rocprofv3 --pmc $c --kernel-trace -d /tmp/prof_b -- $cmd > /tmp/prof_b.log 2>&1
    db=$(find /tmp/prof_b -name "*.db" | head -1)
    [ -n "$db" ] && python $R/profiles/summarize_rocpd.py pmc $db ${OUT}_${which}_$tag.csv
  done
done
for f in ${OUT}_*.csv; do echo "== $f"; grep -v "fill\|copy\|elementwise\|randn\|distribution\|^kernel" $f | awk -F'",' '{n=split($1,a,"::"); print substr(a[n],1,50) "," $2}' | cut -d, -f1-4,7,8-9; done
