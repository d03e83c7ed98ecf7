#!/bin/bash
# same-box A/B of the grouped-query decode kernel: round-5 file against HEAD (32-key pair steps + scalar row bases), fixed shapes and config 4's step
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
cd $GRAFT_REPO_ROOT
python bench.py --model llama2-70b --weight-quant 4 --batch 256 --kv-len 2048 --emulate-tp 8 --no-cpu-baseline --no-serving-leg --no-i8i8-leg --prefill-sample 0 --ragged-steps 0 --breakdown-steps 0 2>/dev/null | python -c "import sys,json; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('  config 4 per rank: ms_per_step', r['ms_per_step'], 'attention', r['breakdown_ms_per_step']['attn_decode'], 'ms', r['roofline']['achieved'], 'GB/s', r['roofline']['frac'])"
cd $GRAFT_REPO_ROOT/ppl.llm.serving_amd/csrc
}
for rep in 1 2; do
  cp $GRAFT_REPO_ROOT/profiles/probes/k_attn_decode_gqa_r05.hip.txt k_attn_decode_gqa.hip; make -s -j16 >/dev/null 2>&1; echo "== round-5 kernel (rep $rep)"; run
  cp /tmp/gq_new.hip k_attn_decode_gqa.hip; make -s -j16 >/dev/null 2>&1; echo "== HEAD: 32-key pair steps (rep $rep)"; run
done
