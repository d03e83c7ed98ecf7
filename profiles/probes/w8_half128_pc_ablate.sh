#!/bin/bash
# NOTE: needs the kernel of profiles/probes/w8_half128_producer_waves_v2.patch (or ..._balanced_k_ranges_v1.patch) applied and built first:
#   git apply profiles/probes/w8_half128_producer_waves_v2.patch && make -C ppl.llm.serving_amd/csrc     (measured, not adopted: profiles/r05_w8_midbatch.md)
# round 5: what bounds gemm_w8_half128_pc_kernel at 64 rows?  Diagnosis builds (WRONG results): 1 = one activation piece per stage instead of
# 16 (the weight stream alone on the LDS-DMA path), 2 = nobody multiplies (the rings stream), 3 = both.  HBM-cold micro-benchmark.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
C=ppl.llm.serving_amd/csrc
mkdir -p gpurun_out; O=gpurun_out/w8_half128_pc_ablate.log; : > $O
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -mllvm -amdgpu-mfma-vgpr-form=1"
OBJS=$(cd $C && ls *.o | grep -v k_gemm_half128.o | sed "s#^#$C/#")
for a in 0 1 2 3; do
  mkdir -p /tmp/ab$a
  /opt/rocm/bin/hipcc $FLAGS -DHP_ABLATE_BUILD=$a -c $C/k_gemm_half128.hip -o /tmp/ab$a/k.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/ab$a/libpplhip.so $OBJS /tmp/ab$a/k.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib || exit 1
  echo "== ablate $a" >> $O
  PPLHIP_LIB=/tmp/ab$a/libpplhip.so python profiles/gemv_microbench.py 8 ${MS:-8 64 128} 2>&1 | grep "^M=" >> $O
done
# the weight stream alone (ablate 3) with stages of (16 KiB / CB) rows x CB contiguous bytes per row: what does the piece shape cost?
for cb in 128 256 512 1024; do
  mkdir -p /tmp/cb$cb
  /opt/rocm/bin/hipcc $FLAGS -DHP_ABLATE_BUILD=3 -DHP_STREAM_CB=$cb -c $C/k_gemm_half128.hip -o /tmp/cb$cb/k.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/cb$cb/libpplhip.so $OBJS /tmp/cb$cb/k.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib || exit 1
  echo "== ablate 3, stage rows x bytes = $((16384 / cb)) x $cb" >> $O
  PPLHIP_LIB=/tmp/cb$cb/libpplhip.so python profiles/gemv_microbench.py 8 8 2>&1 | grep "^M=" >> $O
done
cat $O
