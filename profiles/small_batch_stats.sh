#!/bin/bash
# per-kernel durations of small-batch decode steps (launch-latency regime): bash profiles/small_batch_stats.sh
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
LEAN="--no-cpu-baseline --no-serving-leg --prefill-sample 0 --ragged-steps 0 --steps 32 --warmup 4"
for cfg in "1 8 8" "1 0 0" "8 8 8"; do
  set -- $cfg
  tag=b$1_w$2_kv$3
  rm -rf /tmp/prof_$tag
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -- python $R/bench.py $LEAN --batch $1 --weight-quant $2 --kv-quant $3 > /tmp/prof_$tag.log 2>&1
  tail -1 /tmp/prof_$tag.log | cut -c1-200
  db=$(find /tmp/prof_$tag -name "*.db" | head -1)
  [ -n "$db" ] && python $R/profiles/summarize_rocpd.py stats $db $R/gpurun_out/small_${tag}_kernel_stats.csv && head -16 $R/gpurun_out/small_${tag}_kernel_stats.csv
done
