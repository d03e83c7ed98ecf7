#!/bin/bash
# NOTE: needs the kernel of profiles/probes/w8_half128_producer_waves_v2.patch (or ..._balanced_k_ranges_v1.patch) applied and built first:
#   git apply profiles/probes/w8_half128_producer_waves_v2.patch && make -C ppl.llm.serving_amd/csrc     (measured, not adopted: profiles/r05_w8_midbatch.md)
# round 5: kernel durations (rocprofv3 --kernel-trace --stats) of the layer linears at M = 8 / 64, tile grid (PPLHIP_GEMM_HALF128_PC=0) against
# balanced K ranges (1): what the micro-benchmark's per-call time holds besides the kernel
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out; O=gpurun_out/w8_ranges_kernel_times.log; : > $O
for m in ${MS:-8 64}; do for r in 0 1; do for sh in wqkv wo w13 w2; do
  rm -rf /tmp/rp; SHAPES=$sh PPLHIP_GEMM_HALF128_PC=$r rocprofv3 --kernel-trace --stats -d /tmp/rp -o t -- python profiles/gemv_microbench.py 8 $m > /tmp/rp.log 2>&1
  db=$(find /tmp/rp -name "*.db" | head -1)
  python profiles/summarize_rocpd.py stats $db /tmp/rp_stats.csv > /dev/null 2>&1
  echo "M=$m pc=$r $sh: $(grep "^M=" /tmp/rp.log | sed 's/.*|| layer//') | kernels: $(python3 -c "
import csv
for r in csv.DictReader(open('/tmp/rp_stats.csv')):
    k=r['kernel']
    if 'gemm' in k or 'reduce' in k: print(k.split('(')[0].split('::')[-1], r['avg_us'], end='  ')
")" >> $O
done; done; done
cat $O
