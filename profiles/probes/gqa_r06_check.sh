#!/bin/bash
# grouped-query decode kernel after a change: its operator / model parity cases, then the fixed-shape timings (one box)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py -x -q -k "attention or attn or decode" 2>&1 | tail -3
python -m pytest tests/test_gpu_tp.py -x -q -k "llama70b" 2>&1 | tail -3
python -m pytest tests/test_gpu_config34_shape.py tests/test_gpu_model.py -x -q 2>&1 | tail -3
python - <<'PY'
import sys, os
sys.argv = ["x", "/dev/null"]
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "profiles"))
import roofline_sweep as rs
for B, KV in ((256, 2048), (256, 1024), (512, 1024), (1024, 512), (256, 4096), (64, 2048), (1024, 1024)):
    sp = rs.heuristic_split(B, KV, 8, 1)
    r = rs.run(B, KV, 8, 1, sp)
    print(f"  B {B:5d} kv {KV:5d} split {sp}: {r['us_per_launch']:8.2f} us  {r['GBps']:7.1f} GB/s  {r['frac_of_8TBps']:.3f}")
PY
