#!/bin/bash
# NOTE: needs profiles/probes/w4_interleaved_nibbles_probe.patch applied (the -DPC_INTERLEAVED_NIBBLES conversion)
# round 5: what would a nibble-interleaved device layout of the int4 weights buy gemm_w4_pc_kernel?  (word = k0 k2 k4 k6 | k1 k3 k5 k7 so
# that (w >> 4 i) & 0x000f000f is the pair (k 2i, k 2i + 1): 15 instead of 19 instructions per 8 weights.)  Diagnosis build: the kernel
# converts AS IF the weights were laid out that way (-DPC_INTERLEAVED_NIBBLES; WRONG results on the natural layout), timed on the 70B / TP8
# layer shapes at M = 256 against the product build.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
C=ppl.llm.serving_amd/csrc
mkdir -p /tmp/iv gpurun_out; O=gpurun_out/w4_interleaved_nibbles_probe.log; : > $O
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden"
/opt/rocm/bin/hipcc $FLAGS -DPC_INTERLEAVED_NIBBLES -c $C/k_gemm_pc.hip -o /tmp/iv/k_gemm_pc.o || exit 1
OBJS=$(cd $C && ls *.o | grep -v k_gemm_pc.o | sed "s#^#$C/#")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/iv/libpplhip.so $OBJS /tmp/iv/k_gemm_pc.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib || exit 1
for rep in 1 2; do
  echo "== product (19 instructions per 8 weights)" >> $O
  python profiles/gemm_microbench.py 256 4 70b-tp8 2>&1 | grep -v "^/opt" | tail -6 >> $O
  echo "== as if nibble-interleaved (15 instructions per 8 weights; wrong results)" >> $O
  PPLHIP_LIB=/tmp/iv/libpplhip.so python profiles/gemm_microbench.py 256 4 70b-tp8 2>&1 | grep -v "^/opt" | tail -6 >> $O
done
cat $O
