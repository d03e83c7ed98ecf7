#!/bin/bash
# Round 4: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the M = 64 layer linears with three blocks per CU (round-3 heuristic,
# PPLHIP_GEMM_HALF128_BLOCKS=768) and with one (default): the split-K slabs are what the extra blocks cost.
# usage (GPU box, repo root): bash profiles/probes/half128_slab_traffic.sh
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for nb in 768 256; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/st_${nb}_$c
    PPLHIP_GEMM_HALF128_BLOCKS=$nb timeout 600 rocprofv3 --pmc $c --kernel-trace -d /tmp/st_${nb}_$c -- python $R/profiles/gemv_microbench.py 8 64 > /tmp/st.log 2>&1
    db=$(find /tmp/st_${nb}_$c -name "*.db" | head -1)
    [ -n "$db" ] && python $R/profiles/summarize_rocpd.py pmc $db $R/gpurun_out/r04_half128_slab_${nb}_$c.csv
  done
done
