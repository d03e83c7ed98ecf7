#!/bin/bash
# same-box A/B: pair-step kernel as first committed (v1) against HEAD (fast load path without clamps for whole sub-tiles, one row product per slab,
# probabilities-only masking), then the kernel's parity cases at HEAD
cd $GRAFT_REPO_ROOT/ppl.llm.serving_amd/csrc
cp k_attn_decode_gqa.hip /tmp/gq_new.hip
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
for rep in 1 2; do
  # (v1 = the pair-step kernel as committed: the box has no .git -- save `git show HEAD:ppl.llm.serving_amd/csrc/k_attn_decode_gqa.hip` as
  #  profiles/probes/k_attn_decode_gqa_pair_v1.hip.txt before sending the tree)
  cp $GRAFT_REPO_ROOT/profiles/probes/k_attn_decode_gqa_pair_v1.hip.txt k_attn_decode_gqa.hip; make -s -j16 >/dev/null 2>&1; echo "== pair steps v1 (rep $rep)"; run
  cp /tmp/gq_new.hip k_attn_decode_gqa.hip; make -s -j16 >/dev/null 2>&1; echo "== HEAD (rep $rep)"; run
done
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py -x -q -k "attention or attn or decode" 2>&1 | tail -2
python -m pytest tests/test_gpu_tp.py -x -q -k "llama70b" 2>&1 | tail -2
python -m pytest tests/test_gpu_config34_shape.py -x -q 2>&1 | tail -2
python -m pytest tests/test_gpu_model.py -x -q 2>&1 | grep -E "passed|failed" | tail -2
