#!/bin/bash
# Wire-level run of the published benchmarking procedure (reference README.md:84-90): gRPC server + token-in/token-out load
# generator on the same box.  usage (GPU box, repo root): bash profiles/grpc_wire_bench.sh [request-rate ...]
set -u
R=$GRAFT_REPO_ROOT
PKG=$R/ppl.llm.serving_amd
PORT=23391
python $PKG/serving/grpc_server.py --model-param-path $PKG/configs/llama2_7b_w8a16_kv8_paged.json --synthetic-weights \
    --host 127.0.0.1 --port $PORT 2> $R/gpurun_out/grpc_server.log &
SRV=$!
for i in $(seq 1 300); do grep -q listening $R/gpurun_out/grpc_server.log 2>/dev/null && break; sleep 1; done
grep listening $R/gpurun_out/grpc_server.log || { echo "server did not start"; kill $SRV; exit 1; }
: > $R/gpurun_out/r01_grpc_wire.jsonl
for rate in "${@:-inf}"; do
  python $PKG/serving/client_qps_measure_token_in_out.py --target 127.0.0.1:$PORT --num-requests 1024 --request-rate $rate \
      | tee -a $R/gpurun_out/r01_grpc_wire.jsonl
done
kill $SRV
wait $SRV 2>/dev/null
exit 0
