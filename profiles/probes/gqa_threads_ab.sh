#!/bin/bash
# grouped-query decode attention: waves per block x register buffers (build switches GQ_THREADS_N, GQ_NBUF) at config 4's shape and its neighbours.
# 512 threads = 8 waves = 2 per SIMD with ONE block per CU (a second block does not fit the registers); 768 = 12 waves = 3 per SIMD.
# usage (GPU box, repo root): bash profiles/probes/gqa_threads_ab.sh
cd $GRAFT_REPO_ROOT/ppl.llm.serving_amd/csrc
for v in "512 2" "768 2" "768 1" "768 3" "1024 1" "1024 2"; do
  set -- $v
  make -s -j16 EXTRA="-DGQ_THREADS_N=$1 -DGQ_NBUF=$2" > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  echo "== GQ_THREADS_N=$1 GQ_NBUF=$2"
  python - <<'PY'
import sys, os
sys.argv = ["x", "/dev/null"]
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "profiles"))
import roofline_sweep as rs
for B, KV in ((256, 2048), (256, 1024), (512, 1024), (1024, 512), (256, 4096), (64, 2048)):
    sp = rs.heuristic_split(B, KV, 8, 1)
    r = rs.run(B, KV, 8, 1, sp)
    print(f"  B {B:5d} kv {KV:5d} split {sp}: {r['us_per_launch']:8.2f} us  {r['GBps']:7.1f} GB/s  {r['frac_of_8TBps']:.3f}")
PY
done
make -s -j16 > /dev/null 2>&1
