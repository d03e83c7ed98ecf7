#!/bin/bash
# grouped-query decode kernel with 32-key pair steps: one pair in flight per wave (GQ_NBUF=2) against two (GQ_NBUF=4); same box, interleaved
cd $GRAFT_REPO_ROOT/ppl.llm.serving_amd/csrc
run() {
python - <<'PY'
import sys, os
sys.argv = ["x", "/dev/null"]
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "profiles"))
import roofline_sweep as rs
for B, KV in ((256, 2048), (512, 1024), (1024, 512), (256, 4096), (256, 1024)):
    r = rs.run(B, KV, 8, 1, 1)
    print(f"  B {B:5d} kv {KV:5d}: {r['us_per_launch']:8.2f} us  {r['GBps']:7.1f} GB/s  {r['frac_of_8TBps']:.3f}")
PY
}
for rep in 1 2; do for nb in 2 4; do
  make -s -j16 EXTRA="-DGQ_NBUF=$nb" >/dev/null 2>&1; echo "== GQ_NBUF=$nb (rep $rep)"; run
done; done
make -s -j16 >/dev/null 2>&1
