#!/bin/bash
# Round-6 evidence, collected on the GPU box from the repo root:  bash profiles/collect_r06.sh [part ...]   (default: all parts)
#  bench    the bench.py line, default flags and the driver's flags (--steps 20 --warmup 5)   -> gpurun_out/r06_bench{,_driver_flags}.json
#  stats    rocprofv3 --kernel-trace --stats of the HEADLINE steps only (--ragged-steps 0 --prefill-sample 0 --breakdown-steps 0: every
#           launch of the dominant kernel in the file is a uniform batch-1024 launch, so roofline.frac can be recomputed from avg_us alone)
#           -> gpurun_out/r06_kernel_stats.csv
#  pmc      PMC passes FETCH_SIZE / WRITE_SIZE (separate runs, --kernel-trace only) of `bench.py --steps 24 --warmup 3`
#           -> r06_attn_decode_pmc_{fetch,write}.csv, attn_decode_traffic.json (per kv length, with mean_duration_us)
#  sq       SQ counters of the GEMMs (MFMA busy)                                              -> gpurun_out/r06_gemm_w8_pmc.csv
#  sweep    SURVEY D2 fixed-shape decode-attention sweep                                      -> gpurun_out/r06_roofline_sweep.json
#  config5  benchmark_prefix_cache_offline, 64 x 8192-token prompts sharing 6144 tokens      -> gpurun_out/r06_prefix_cache_benchmark.log
#  small    decode step latency at batch 1 .. 128                                             -> r06_small_batch_latency.txt
#  tp8      bench.py --gpus 8 with all eight ranks on the one device (launch path only)      -> r06_bench_tp8_one_device.json
#  gemm     layer GEMM micro-benchmarks at M = 1024 / 8192 (W8A16), the TP-4 / TP-8 slices at M = 1024, 70B/TP8 at M = 256 (W4A16)
#           -> r06_gemm_microbench.txt; the vendor library on the same shapes the same minute -> r06_hipblaslt_reference.txt
#  w4       config 4: per-rank step with its kernel stats                                    -> r06_config4_kernel_stats.csv, r06_config4_step.log
#  tp       per-rank compute of the tensor-parallel configurations, identity collectives      -> gpurun_out/r06_tp_emulation.txt
#  tpslice  rocprofv3 --kernel-trace --stats of `bench.py --emulate-tp {4,8}` (one rank's slice of the headline step, per kernel, per launch)
#           + SQ counters of the slice's GEMMs -> r06_tp{4,8}_slice_kernel_stats.csv, r06_tp8_slice_sq.csv
# Summaries are copied into profiles/ by hand (gpurun_out/ is scratch).
set -u
R=$GRAFT_REPO_ROOT
PARTS=${*:-bench stats pmc sq sweep config5 tp small tp8 gemm w4 tpslice}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
LEAN="--no-cpu-baseline --no-serving-leg --no-i8i8-leg --prefill-sample 0"
PURE="$LEAN --ragged-steps 0 --breakdown-steps 0"
for part in $PARTS; do case $part in
bench)
  python $R/bench.py 2>$R/gpurun_out/r06_bench.err > $R/gpurun_out/r06_bench.json
  python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-serving-leg --no-i8i8-leg 2>/dev/null > $R/gpurun_out/r06_bench_driver_flags.json
  python $R/bench.py --breakdown $LEAN --ragged-steps 0 2>/dev/null > $R/gpurun_out/r06_bench_breakdown.json ;;
stats)
  rm -rf /tmp/prof_stats
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -- python $R/bench.py $PURE > /tmp/prof_stats.log 2>&1
  db=$(find /tmp/prof_stats -name "*.db" | head -1)
  [ -n "$db" ] && python $R/profiles/summarize_rocpd.py stats $db $R/gpurun_out/r06_kernel_stats.csv ;;
pmc)
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/prof_$c
    timeout 1200 rocprofv3 --pmc $c --kernel-trace -d /tmp/prof_$c -- python $R/bench.py $PURE --steps 24 --warmup 3 > /tmp/prof_$c.log 2>&1
  done
  dbf=$(find /tmp/prof_FETCH_SIZE -name "*.db" | head -1); dbw=$(find /tmp/prof_WRITE_SIZE -name "*.db" | head -1)
  [ -n "$dbf" ] && python $R/profiles/summarize_rocpd.py pmc $dbf $R/gpurun_out/r06_attn_decode_pmc_fetch.csv
  [ -n "$dbw" ] && python $R/profiles/summarize_rocpd.py pmc $dbw $R/gpurun_out/r06_attn_decode_pmc_write.csv
  # kv_len 512, steps 0..26 (3 warm-up + 24 timed): step i launches the kernel 32 times at kv length 512 + i + 1
  [ -n "$dbf" ] && [ -n "$dbw" ] && python $R/profiles/summarize_rocpd.py traffic_table $dbf $dbw $R/gpurun_out/attn_decode_traffic.json llama2-7b 1024 512 32 27 "HEAD round 6" ;;
sq)
  rm -rf /tmp/prof_sq
  timeout 1200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 --kernel-trace -d /tmp/prof_sq -- python $R/bench.py $PURE --steps 2 --warmup 1 > /tmp/prof_sq.log 2>&1
  db=$(find /tmp/prof_sq -name "*.db" | head -1)
  [ -n "$db" ] && python $R/profiles/summarize_rocpd.py pmc $db $R/gpurun_out/r06_gemm_w8_pmc.csv ;;
sweep)
  python $R/profiles/roofline_sweep.py $R/gpurun_out/r06_roofline_sweep.json > $R/gpurun_out/r06_roofline_sweep.log 2>&1 ;;
config5)
  cd $R/ppl.llm.serving_amd
  ./build/benchmark_prefix_cache_offline --model-param-path configs/llama2_7b_w8a16_kv8_paged.json --synthetic-weights --enable-prefix-cache \
     --max-prefill-batch 1 --max-input-tokens-per-request 8192 --max-total-tokens-per-request 16384 --batch 64 > $R/gpurun_out/r06_prefix_cache_benchmark.log 2>&1
  ./build/benchmark_prefix_cache_offline --model-param-path configs/llama2_7b_w8a16_kv8_paged.json --synthetic-weights --enable-prefix-cache \
     --max-prefill-batch 1 --max-input-tokens-per-request 8192 --max-total-tokens-per-request 16384 --batch 1 --second-run new-tails >> $R/gpurun_out/r06_prefix_cache_benchmark.log 2>&1
  cd /tmp ;;
tp)
  OUT=$R/gpurun_out/r06_tp_emulation.txt
  L="$LEAN --ragged-steps 0 --breakdown"
  fmt='import sys,json; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(sys.argv[1], r["ms_per_step"], r["breakdown_ms_per_step"], r["roofline"]["achieved"], r["roofline"]["kernel"])'
  echo "# bench.py --emulate-tp N $L  (one rank's slice, identity collectives; columns: ms/step, breakdown, decode-attention algorithmic GB/s)" > $OUT
  for tp in 2 4 8; do python $R/bench.py --emulate-tp $tp $L 2>/dev/null | python -c "$fmt" "7b_w8a16_b1024_kv512_tp$tp" >> $OUT; done
  python $R/bench.py --model llama2-13b --batch 512 --kv-len 1024 --emulate-tp 2 $L 2>/dev/null | python -c "$fmt" "13b_w8a16_b512_kv1024_tp2(config3)" >> $OUT
  python $R/bench.py --model llama2-70b --weight-quant 4 --batch 256 --kv-len 2048 --emulate-tp 8 $L 2>/dev/null | python -c "$fmt" "70b_w4a16_b256_kv2048_tp8(config4)" >> $OUT
  fmt2='import sys,json; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(sys.argv[1], "ms_per_step", r["ms_per_step"], "attn", r["breakdown_ms_per_step"]["attn_decode"], r["roofline"]["achieved"], "GB/s")'
  echo "# without --breakdown (only the decode-attention launches carry their dispatch-packet timestamps):" >> $OUT
  for tp in 2 4 8; do python $R/bench.py --emulate-tp $tp $PURE 2>/dev/null | python -c "$fmt2" "7b_w8a16_b1024_kv512_tp$tp" >> $OUT; done
  python $R/bench.py --model llama2-13b --batch 512 --kv-len 1024 --emulate-tp 2 $PURE 2>/dev/null | python -c "$fmt2" "13b_w8a16_b512_kv1024_tp2(config3)" >> $OUT
  python $R/bench.py --model llama2-70b --weight-quant 4 --batch 256 --kv-len 2048 --emulate-tp 8 $PURE 2>/dev/null | python -c "$fmt2" "70b_w4a16_b256_kv2048_tp8(config4)" >> $OUT ;;
tpslice)
  for tp in 4 8; do
    rm -rf /tmp/prof_tp$tp
    timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_tp$tp -- python $R/bench.py --emulate-tp $tp $PURE --steps 8 --warmup 2 > /tmp/prof_tp$tp.log 2>&1
    db=$(find /tmp/prof_tp$tp -name "*.db" | head -1)
    [ -n "$db" ] && python $R/profiles/summarize_rocpd.py stats $db $R/gpurun_out/r06_tp${tp}_slice_kernel_stats.csv
  done
  rm -rf /tmp/prof_tp8sq
  timeout 900 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 --kernel-trace -d /tmp/prof_tp8sq -- python $R/bench.py --emulate-tp 8 $PURE --steps 2 --warmup 1 > /tmp/prof_tp8sq.log 2>&1
  db=$(find /tmp/prof_tp8sq -name "*.db" | head -1)
  [ -n "$db" ] && python $R/profiles/summarize_rocpd.py pmc $db $R/gpurun_out/r06_tp8_slice_sq.csv ;;
w4)
  cd $R && STATS=r06_config4_kernel_stats.csv bash profiles/probes/config4_step.sh > gpurun_out/r06_config4_step.log 2>&1; cd /tmp ;;
small)
  python $R/profiles/small_batch_latency.py 1 2 4 8 16 32 64 128 2>&1 | grep batch > $R/gpurun_out/r06_small_batch_latency.txt ;;
tp8)
  cd $R && PPLHIP_COMM=p2p PPLHIP_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus 8 --steps 4 --warmup 1 --no-cpu-baseline --ragged-steps 0 --prefill-sample 0 \
     2>$R/gpurun_out/r06_bench_tp8_one_device.err > $R/gpurun_out/r06_bench_tp8_one_device.json; cd /tmp ;;
gemm)
  OUT=$R/gpurun_out/r06_gemm_microbench.txt; : > $OUT
  python $R/profiles/gemm_microbench.py 1024 8 7b >> $OUT 2>&1
  python $R/profiles/gemm_microbench.py 8192 8 7b >> $OUT 2>&1
  python $R/profiles/gemm_microbench.py 1024 8 7b-tp4 >> $OUT 2>&1
  python $R/profiles/gemm_microbench.py 1024 8 7b-tp8 >> $OUT 2>&1
  python $R/profiles/gemm_microbench.py 256 4 70b-tp8 >> $OUT 2>&1
  python $R/profiles/hipblaslt_reference.py > $R/gpurun_out/r06_hipblaslt_reference.txt 2>&1 ;;
esac; done
ls -la $R/gpurun_out | tail -25
