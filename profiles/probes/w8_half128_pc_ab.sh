#!/bin/bash
# NOTE: needs the kernel of profiles/probes/w8_half128_producer_waves_v2.patch (or ..._balanced_k_ranges_v1.patch) applied and built first:
#   git apply profiles/probes/w8_half128_producer_waves_v2.patch && make -C ppl.llm.serving_amd/csrc     (measured, not adopted: profiles/r05_w8_midbatch.md)
# Round 5: gemm_w8_half128_pc_kernel -- parity, then the HBM-cold micro-benchmark and the decode step A/B against the four-wave kernel.
cd /root/repo; mkdir -p gpurun_out; O=gpurun_out/w8_half128_pc_ab.log; : > $O
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "producer_waves or half_height or half128 or random_shapes_w8" 2>&1 | tail -8 >> $O
for r in 0 1; do
  echo "== PPLHIP_GEMM_HALF128_PC=$r" >> $O
  PPLHIP_GEMM_HALF128_PC=$r timeout 600 python profiles/gemv_microbench.py 8 8 16 32 64 96 128 >> $O 2>&1
  PPLHIP_GEMM_HALF128_PC=$r timeout 600 python profiles/small_batch_latency.py 8 16 32 64 96 128 >> $O 2>&1
done
cat $O
