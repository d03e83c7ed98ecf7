#!/bin/bash
# Round-1 evidence, collected on the GPU box from the repo root:  bash profiles/collect_r01.sh
# 1. bench.py line  2. rocprofv3 --kernel-trace --stats of the same command (short)  3. PMC passes (FETCH_SIZE, WRITE_SIZE)
#    for the decode attention kernel.  Summaries land in gpurun_out/ and are copied into profiles/ by hand.
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python $R/bench.py 2>$R/gpurun_out/r01_bench.err > $R/gpurun_out/r01_bench.json
rm -rf /tmp/prof_stats
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /tmp/prof_stats.log 2>&1
db=$(find /tmp/prof_stats -name "*.db" | head -1)
[ -n "$db" ] && python $R/profiles/summarize_rocpd.py stats $db $R/gpurun_out/r01_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_$c
  timeout 900 rocprofv3 --pmc $c --kernel-trace -d /tmp/prof_$c -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --prefill-sample 0 > /tmp/prof_$c.log 2>&1
  db=$(find /tmp/prof_$c -name "*.db" | head -1)
  [ -n "$db" ] && python $R/profiles/summarize_rocpd.py pmc $db $R/gpurun_out/r01_pmc_$c.csv
done
ls -la $R/gpurun_out
