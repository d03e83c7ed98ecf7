#!/bin/bash
# Round 4: what does the two-chunk (overlapped) schedule COST a pure decode step of 1024 rows?  One rank's slice with identity collectives
# (bench.py --emulate-tp): the hand-offs and the half-size launches are all there, the all-reduce time they would hide is not.
# usage (GPU box, repo root): bash profiles/probes/tp_decode_overlap_cost.sh
L2="--no-cpu-baseline --no-serving-leg --no-i8i8-leg --prefill-sample 0 --ragged-steps 0 --breakdown-steps 0"
fmt2='import sys,json; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(sys.argv[1], "ms_per_step", r["ms_per_step"], "attn", r["breakdown_ms_per_step"]["attn_decode"], r["roofline"]["achieved"], "GB/s")'
for tp in 8 4 2; do
  python bench.py --emulate-tp $tp $L2 2>/dev/null | python -c "$fmt2" "tp$tp one chunk"
  PPLHIP_TP_OVERLAP_MIN_TOKENS=1024 python bench.py --emulate-tp $tp $L2 2>/dev/null | python -c "$fmt2" "tp$tp two chunks (flags)"
  PPLHIP_TP_HANDOFF=events PPLHIP_TP_OVERLAP_MIN_TOKENS=1024 python bench.py --emulate-tp $tp $L2 2>/dev/null | python -c "$fmt2" "tp$tp two chunks (events)"
done
