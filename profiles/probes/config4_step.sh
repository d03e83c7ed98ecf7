#!/bin/bash
# round 5: config 4 per-rank step (70B W4A16 / TP8 slice, batch 256, kv 2048), A/B of PPLHIP_GEMM_PC, then rocprofv3 kernel stats of the default
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
L="--steps 12 --warmup 4 --no-cpu-baseline --no-serving-leg --no-i8i8-leg --ragged-steps 0 --prefill-sample 0 --breakdown-steps 0"
fmt='import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); print(sys.argv[1], "ms/step", d["ms_per_step"], "attn_frac", d.get("roofline", {}).get("frac"))'
for pc in ${PCS:-0 1 0 1}; do
  PPLHIP_GEMM_PC=$pc ${EXTRA_ENV:-env} python bench.py --model llama2-70b --weight-quant 4 --batch 256 --kv-len 2048 --emulate-tp 8 $L 2>/dev/null | python -c "$fmt" "PC=$pc"
done
if [ -n "$STATS" ]; then
  cd /tmp && rm -rf /tmp/c4prof
  rocprofv3 --kernel-trace --stats -d /tmp/c4prof -- python $GRAFT_REPO_ROOT/bench.py --model llama2-70b --weight-quant 4 --batch 256 --kv-len 2048 --emulate-tp 8 $L > /tmp/c4prof.log 2>&1
  db=$(find /tmp/c4prof -name "*.db" | head -1)
  python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py stats $db $GRAFT_REPO_ROOT/gpurun_out/$STATS
  head -14 $GRAFT_REPO_ROOT/gpurun_out/$STATS | cut -c1-150
fi
