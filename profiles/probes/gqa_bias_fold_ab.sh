#!/bin/bash
# grouped-query decode kernel: the conversion's -1152 bias of K folded into one extra MFMA per sub-tile (GQ_K_BIAS_FOLD=1) against the plain
# conversion (0); same box, interleaved; then parity at the folded form
cd $GRAFT_REPO_ROOT/ppl.llm.serving_amd/csrc
run() {
python - <<'PY'
import sys, os
sys.argv = ["x", "/dev/null"]
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "profiles"))
import roofline_sweep as rs
for B, KV in ((256, 2048), (512, 1024), (1024, 512), (256, 4096), (1024, 2048)):
    sp = rs.heuristic_split(B, KV, 8, 1)
    r = rs.run(B, KV, 8, 1, sp)
    print(f"  B {B:5d} kv {KV:5d}: {r['us_per_launch']:8.2f} us  {r['GBps']:7.1f} GB/s  {r['frac_of_8TBps']:.3f}")
PY
}
for rep in 1 2; do for f in 0 1; do
  make -s -j16 EXTRA="-DGQ_K_BIAS_FOLD=$f" >/dev/null 2>&1; echo "== GQ_K_BIAS_FOLD=$f (rep $rep)"; run
done; done
make -s -j16 >/dev/null 2>&1
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py -x -q -k "attention or attn or decode" 2>&1 | grep -E "passed|failed"
python -m pytest tests/test_gpu_tp.py -x -q -k "llama70b" 2>&1 | grep -E "passed|failed"
python -m pytest tests/test_gpu_config34_shape.py tests/test_gpu_model.py -x -q 2>&1 | grep -E "passed|failed"
grep -E "llama70b|hf_gqa|gqa" gpurun_out/parity_errors.jsonl | tail -8
