#!/bin/bash
# per-rank compute of tensor-parallel decode steps with identity collectives (one MI355X): bash profiles/tp_emulation_r02.sh
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_tp_emulation.txt
L="--no-cpu-baseline --no-serving-leg --no-i8i8-leg --prefill-sample 0 --ragged-steps 0 --breakdown"
fmt='import sys,json; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(sys.argv[1], r["ms_per_step"], r["breakdown_ms_per_step"], r["roofline"]["achieved"], r["roofline"]["kernel"])'
echo "# bench.py --emulate-tp N $L  (one rank's slice, identity collectives; columns: ms/step, breakdown, decode-attention algorithmic GB/s)" > $OUT
for tp in 2 4 8; do python $R/bench.py --emulate-tp $tp $L 2>/dev/null | python -c "$fmt" "7b_w8a16_b1024_kv512_tp$tp" >> $OUT; done
python $R/bench.py --model llama2-13b --batch 512 --kv-len 1024 --emulate-tp 2 $L 2>/dev/null | python -c "$fmt" "13b_w8a16_b512_kv1024_tp2(config3)" >> $OUT
python $R/bench.py --model llama2-70b --weight-quant 4 --batch 256 --kv-len 2048 --emulate-tp 8 $L 2>/dev/null | python -c "$fmt" "70b_w4a16_b256_kv2048_tp8(config4)" >> $OUT
python $R/bench.py --cache-mode 1 $L 2>/dev/null | python -c "$fmt" "7b_w8a16_b1024_kv512_tp1_paged16" >> $OUT
cat $OUT
