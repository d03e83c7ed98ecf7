#!/bin/bash
# Round-2 evidence, collected on the GPU box from the repo root:  bash profiles/collect_r02.sh [label]
#  1. the bench.py line (default flags = what the driver runs)              -> gpurun_out/r02_bench.json
#  2. rocprofv3 --kernel-trace --stats of the same command                   -> gpurun_out/r02_kernel_stats.csv
#  3. PMC passes FETCH_SIZE / WRITE_SIZE (separate runs, --kernel-trace only) of the same decode steps
#                                                                            -> gpurun_out/r02_attn_decode_pmc_{fetch,write}.csv
#                                                                               gpurun_out/attn_decode_traffic.json
#  4. SQ counters of the GEMMs (MFMA busy)                                   -> gpurun_out/r02_gemm_w8_pmc.csv
# Summaries are copied into profiles/ by hand (gpurun_out/ is scratch).
set -u
R=$GRAFT_REPO_ROOT
LABEL=${1:-r02}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python $R/bench.py 2>$R/gpurun_out/r02_bench.err > $R/gpurun_out/r02_bench.json
# the same steps with every kernel class bracketed by events (GEMM ms/step; costs the step ~3 %)
python $R/bench.py --breakdown --no-cpu-baseline --no-serving-leg --no-i8i8-leg --ragged-steps 0 2>/dev/null > $R/gpurun_out/r02_bench_breakdown.json
LEAN="--no-cpu-baseline --no-serving-leg --no-i8i8-leg --prefill-sample 0"
rm -rf /tmp/prof_stats
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -- python $R/bench.py $LEAN > /tmp/prof_stats.log 2>&1
db=$(find /tmp/prof_stats -name "*.db" | head -1)
[ -n "$db" ] && python $R/profiles/summarize_rocpd.py stats $db $R/gpurun_out/r02_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_$c
  timeout 1200 rocprofv3 --pmc $c --kernel-trace -d /tmp/prof_$c -- python $R/bench.py $LEAN --ragged-steps 0 > /tmp/prof_$c.log 2>&1
done
dbf=$(find /tmp/prof_FETCH_SIZE -name "*.db" | head -1); dbw=$(find /tmp/prof_WRITE_SIZE -name "*.db" | head -1)
[ -n "$dbf" ] && python $R/profiles/summarize_rocpd.py pmc $dbf $R/gpurun_out/r02_attn_decode_pmc_fetch.csv
[ -n "$dbw" ] && python $R/profiles/summarize_rocpd.py pmc $dbw $R/gpurun_out/r02_attn_decode_pmc_write.csv
# default bench: batch 1024, kv_len 512, warm-up 3, 16 steps -> timed launches at kv 516 .. 531
[ -n "$dbf" ] && [ -n "$dbw" ] && python $R/profiles/summarize_rocpd.py traffic $dbf $dbw $R/gpurun_out/attn_decode_traffic.json 1024 516 531 32 3 16 "$LABEL"
rm -rf /tmp/prof_sq
timeout 1200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 --kernel-trace -d /tmp/prof_sq -- python $R/bench.py $LEAN --ragged-steps 0 --steps 2 --warmup 1 > /tmp/prof_sq.log 2>&1
db=$(find /tmp/prof_sq -name "*.db" | head -1)
[ -n "$db" ] && python $R/profiles/summarize_rocpd.py pmc $db $R/gpurun_out/r02_gemm_w8_pmc.csv
ls -la $R/gpurun_out | tail -20
