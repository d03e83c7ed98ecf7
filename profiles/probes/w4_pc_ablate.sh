#!/bin/bash
# round 5: what bounds gemm_w4_pc_kernel?  Diagnosis build (-DPC_ABLATE_BUILD, WRONG results) of k_gemm_pc.hip linked into a second library;
# PPLHIP_PC_ABL bits: 1 no MFMAs, 2 no conversion, 4 no activation refills, 8 no raw-weight refills, 16 no fragment reads
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
C=ppl.llm.serving_amd/csrc
mkdir -p /tmp/abl
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden"
/opt/rocm/bin/hipcc $FLAGS -DPC_ABLATE_BUILD ${PC_EXTRA:-} -c $C/k_gemm_pc.hip -o /tmp/abl/k_gemm_pc.o || exit 1
OBJS=$(cd $C && ls *.o | grep -v k_gemm_pc.o | sed "s#^#$C/#")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/abl/libpplhip.so $OBJS /tmp/abl/k_gemm_pc.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib || exit 1
export PPLHIP_LIB=/tmp/abl/libpplhip.so
for sh in ${SHAPES:-w13 w2}; do
for abl in ${ABLS:-0 1 2 4 8 16 3 6 12 14 17 31 0}; do
  echo -n "abl=$abl  "
  PPLHIP_PC_ABL=$abl timeout 300 python profiles/gemm_microbench.py ${MROWS:-256} 4 70b-tp8 $sh 2>&1 | grep "^$sh"
done; done
