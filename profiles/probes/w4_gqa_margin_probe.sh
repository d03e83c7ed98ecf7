#!/bin/bash
# round 5: where does llama70b_tp8_w4a16_decode_kv2048 spend its parity margin?  The same test with V exact / rounded in the
# grouped-query decode kernel (a second library built with -DGQ_V_EXACT=1) x RMSNorm on 256 / 1024 threads; error and noise floor per variant.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
C=ppl.llm.serving_amd/csrc
mkdir -p /tmp/vx
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden"
/opt/rocm/bin/hipcc $FLAGS -DGQ_V_EXACT=1 -c $C/k_attn_decode_gqa.hip -o /tmp/vx/k_attn_decode_gqa.o || exit 1
OBJS=$(cd $C && ls *.o | grep -v k_attn_decode_gqa.o | sed "s#^#$C/#")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/vx/libpplhip.so $OBJS /tmp/vx/k_attn_decode_gqa.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib || exit 1
for lib in default vexact; do
  for wide in 0 512; do
    log=/tmp/par_${lib}_$wide.jsonl; rm -f $log
    if [ $lib = vexact ]; then export PPLHIP_LIB=/tmp/vx/libpplhip.so; else unset PPLHIP_LIB; fi
    PPLHIP_RMSNORM_WIDE_MAX_ROWS=$wide PPLHIP_PARITY_LOG=$log timeout 900 python -m pytest tests/test_gpu_tp.py -q -m gpu -k "llama70b" 2>&1 | tail -2 | head -1
    echo "lib=$lib rmsnorm_wide_max_rows=$wide: $(grep decode_kv2048 $log)"
  done
done
