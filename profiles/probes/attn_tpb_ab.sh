#!/bin/bash
# round 5: decode attention with 128 / 256 / 512 threads per workgroup (PPLHIP_ATTN_TPB, a tuning switch: needs a -DPPLHIP_TUNING_BUILD
# object of k_attn_decode.hip) on the headline step (kv ~520) and the ragged leg (kv 7 .. 1023, mean 248)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
C=ppl.llm.serving_amd/csrc
mkdir -p /tmp/tpb gpurun_out; O=gpurun_out/attn_tpb_ab.log; : > $O
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden"
/opt/rocm/bin/hipcc $FLAGS -DPPLHIP_TUNING_BUILD -c $C/k_attn_decode.hip -o /tmp/tpb/k_attn_decode.o || exit 1
OBJS=$(cd $C && ls *.o | grep -v k_attn_decode.o | sed "s#^#$C/#")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/tpb/libpplhip.so $OBJS /tmp/tpb/k_attn_decode.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib || exit 1
for rep in 1 2; do for t in 128 256 512; do
  PPLHIP_LIB=/tmp/tpb/libpplhip.so PPLHIP_ATTN_TPB=$t python bench.py --steps 10 --warmup 3 --no-serving-leg --no-i8i8-leg --no-cpu-baseline --ragged-steps 6 > /tmp/b.json 2>/tmp/b.err
  python3 - $t >> $O <<'PY'
import json, sys
r = json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
rb = r.get("ragged_batch", {})
print(f"threads {sys.argv[1]}: headline {r['ms_per_step']} ms/step, attention frac {r['roofline']['frac']}; ragged {rb.get('ms_per_step')} ms/step, attention frac {rb.get('attn_decode_frac_of_8TBps')}")
PY
done; done
cat $O
