#!/bin/bash
# grouped-query decode kernel: 8-wave blocks (one per CU) against 4-wave blocks (two per CU) by launch size; PPLHIP_GQA_SMALL_BLOCK_MIN = number of
# blocks from which the 4-wave form is used (default 512).  Same box, interleaved.
cd $GRAFT_REPO_ROOT
run() {
python - <<'PY'
import sys, os
sys.argv = ["x", "/dev/null"]
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "profiles"))
import roofline_sweep as rs
for B, KV in ((256, 2048), (256, 1024), (512, 1024), (512, 2048), (1024, 512), (1024, 1024), (1024, 2048), (256, 4096), (64, 2048)):
    sp = rs.heuristic_split(B, KV, 8, 1)
    r = rs.run(B, KV, 8, 1, sp)
    print(f"  B {B:5d} kv {KV:5d} split {sp}: {r['us_per_launch']:8.2f} us  {r['GBps']:7.1f} GB/s  {r['frac_of_8TBps']:.3f}")
PY
}
for rep in 1 2; do for m in 1000000000 1; do echo "== PPLHIP_GQA_SMALL_BLOCK_MIN=$m (rep $rep)"; PPLHIP_GQA_SMALL_BLOCK_MIN=$m run; done; done
python -m pytest tests/test_gpu_ops.py -x -q -k "attention or attn or decode" 2>&1 | tail -2
PPLHIP_GQA_SMALL_BLOCK_MIN=1 python -m pytest tests/test_gpu_ops.py tests/test_gpu_config34_shape.py -x -q -k "attention or attn or decode" 2>&1 | tail -2
PPLHIP_GQA_SMALL_BLOCK_MIN=1 python -m pytest tests/test_gpu_tp.py -x -q -k "llama70b" 2>&1 | tail -2
