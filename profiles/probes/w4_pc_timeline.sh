#!/bin/bash
# round 5: cycle stamps of every wave of block 0 of gemm_w4_pc_kernel at the protocol's points (-DPC_TIME_BUILD; PPLHIP_PC_TIME=1 prints the
# fifth call's stamps relative to each wave's start): where does a phase's time go?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
C=ppl.llm.serving_amd/csrc
mkdir -p /tmp/pt
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden"
/opt/rocm/bin/hipcc $FLAGS -DPC_TIME_BUILD -c $C/k_gemm_pc.hip -o /tmp/pt/k_gemm_pc.o || exit 1
OBJS=$(cd $C && ls *.o | grep -v k_gemm_pc.o | sed "s#^#$C/#")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/pt/libpplhip.so $OBJS /tmp/pt/k_gemm_pc.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib || exit 1
for sh in ${SHAPES:-w13 wo}; do
  PPLHIP_LIB=/tmp/pt/libpplhip.so PPLHIP_PC_TIME=1 python profiles/gemm_microbench.py 256 4 70b-tp8 $sh 2>&1 | grep "pc_time\|^$sh"
done
