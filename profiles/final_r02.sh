#!/bin/bash
# last evidence pass of round 2 on the GPU box: bash profiles/final_r02.sh
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
( time python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err ) 2> gpurun_out/r02_bench_time.txt
python bench.py --breakdown --no-cpu-baseline --no-serving-leg --no-i8i8-leg --ragged-steps 0 2>/dev/null > gpurun_out/r02_bench_breakdown.json
cd $R/ppl.llm.serving_amd
./build/benchmark_prefix_cache_offline --model-param-path configs/llama2_7b_w8a16_kv8_paged.json --synthetic-weights --enable-prefix-cache \
   --max-prefill-batch 1 --max-input-tokens-per-request 8192 --max-total-tokens-per-request 16384 > $R/gpurun_out/r02_prefix_cache_benchmark.log 2>&1
./build/offline_inference --model-param-path configs/llama2_7b_w8a16_kv8_paged.json --synthetic-weights --workload samples1024 2>/dev/null | tail -1 > $R/gpurun_out/r02_serving_samples1024_paged.json
cd $R
tail -c 1500 gpurun_out/r02_bench.json; echo; cat gpurun_out/r02_bench_time.txt; tail -6 gpurun_out/r02_prefix_cache_benchmark.log; cat gpurun_out/r02_serving_samples1024_paged.json | cut -c1-600
