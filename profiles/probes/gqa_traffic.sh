#!/bin/bash
# HBM traffic of the grouped-query decode kernel at config 4's shape (B 256, kv 2048, 8 query heads on 1 KV head, int8-g8 KV): FETCH_SIZE
# and TCP / TCC request counters against the algorithmic bytes (168.8 MB per launch) -- are the half-line K loads fetched twice?
# usage (GPU box, repo root): bash profiles/probes/gqa_traffic.sh <out-prefix>
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-gqa_traffic}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/gqa_one.py <<'PY'
import sys, os
sys.argv = ["x", "/dev/null"]
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "profiles"))
import roofline_sweep as rs
for B, KV in ((256, 2048), (1024, 512)):
    r = rs.run(B, KV, 8, 1, 1, iters=16, warm=4)
    print(r)
PY
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  tag=$(echo $c | tr ' ' '_')
  rm -rf /tmp/prof_g
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d /tmp/prof_g -- python /tmp/gqa_one.py > /tmp/prof_g.log 2>&1
  db=$(find /tmp/prof_g -name "*.db" | head -1)
  [ -n "$db" ] && python $R/profiles/summarize_rocpd.py pmc $db ${OUT}_$tag.csv || tail -5 /tmp/prof_g.log
done
grep -h attn_decode_gqa ${OUT}_*.csv | cut -d, -f2- | cut -c1-400
