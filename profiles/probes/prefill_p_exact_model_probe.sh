#!/bin/bash
# round 5: model-level effect of the prefill kernel's P term rounded once (-DP3_P_EXACT=0) against the exact hi + lo pair:
# the parity logs of the prefill-heavy model tests with both libraries, then the 8192-token operator time of both.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
C=ppl.llm.serving_amd/csrc
mkdir -p /tmp/px
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden"
/opt/rocm/bin/hipcc $FLAGS -DP3_P_EXACT=0 -c $C/k_attn_prefill32.hip -o /tmp/px/k_attn_prefill32.o || exit 1
OBJS=$(cd $C && ls *.o | grep -v k_attn_prefill32.o | sed "s#^#$C/#")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/px/libpplhip.so $OBJS /tmp/px/k_attn_prefill32.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib || exit 1
for lib in exact rounded; do
  if [ $lib = rounded ]; then export PPLHIP_LIB=/tmp/px/libpplhip.so; else unset PPLHIP_LIB; fi
  log=$GRAFT_REPO_ROOT/gpurun_out/parity_prefill_p_$lib.jsonl; rm -f $log
  PPLHIP_PARITY_LOG=$log timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_fulldepth.py tests/test_gpu_config34_shape.py tests/test_gpu_config5_tokens.py -q -m gpu -k "not two_stream and not split_k_slabs and not small_batch" 2>&1 | tail -2
  python profiles/attn_prefill_microbench.py 2>&1 | grep -v amdgpu | head -8
done
