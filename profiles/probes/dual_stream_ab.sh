#!/bin/bash
# Two-stream decode (PPLHIP_DUAL_STREAM=1, pplhip.cc run_launches) against the one-stream step: config 4's per-rank step (70B W4A16 / TP8
# slice, identity collectives), 7B/TP8 and 13B/TP2 slices, and the 7B TP=1 step at batch 64..512.   usage: dual_stream_ab.sh [out]
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$R/gpurun_out/r04_dual_stream_ab.txt}
L2="--no-cpu-baseline --no-serving-leg --no-i8i8-leg --prefill-sample 0 --ragged-steps 0 --breakdown-steps 0"
fmt2='import sys,json; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(sys.argv[1], "ms_per_step", r["ms_per_step"], "attn", r["breakdown_ms_per_step"]["attn_decode"], r["roofline"]["achieved"], "GB/s")'
echo "# bench.py $L2 ; dual = PPLHIP_DUAL_STREAM=1 PPLHIP_DUAL_MAX_ROWS=1024" > $OUT
for dual in 0 1 0 1; do
  export PPLHIP_DUAL_STREAM=$dual PPLHIP_DUAL_MAX_ROWS=1024
  python $R/bench.py --model llama2-70b --weight-quant 4 --batch 256 --kv-len 2048 --emulate-tp 8 $L2 2>/dev/null | python -c "$fmt2" "dual=$dual 70b_w4a16_b256_kv2048_tp8(config4)" >> $OUT
done
for dual in 0 1; do
  export PPLHIP_DUAL_STREAM=$dual PPLHIP_DUAL_MAX_ROWS=1024
  python $R/bench.py --model llama2-13b --batch 512 --kv-len 1024 --emulate-tp 2 $L2 2>/dev/null | python -c "$fmt2" "dual=$dual 13b_w8a16_b512_kv1024_tp2(config3)" >> $OUT
  python $R/bench.py --emulate-tp 8 $L2 2>/dev/null | python -c "$fmt2" "dual=$dual 7b_w8a16_b1024_kv512_tp8" >> $OUT
done
echo "# profiles/small_batch_latency.py (7B W8A16 TP=1, kv 512), PPLHIP_DUAL_MIN_ROWS=32" >> $OUT
for dual in 0 1 0 1; do
  PPLHIP_DUAL_STREAM=$dual PPLHIP_DUAL_MIN_ROWS=32 PPLHIP_DUAL_MAX_ROWS=1024 python $R/profiles/small_batch_latency.py 32 64 96 128 192 256 384 512 2>&1 | grep batch | sed "s/^/dual=$dual /" >> $OUT
done
