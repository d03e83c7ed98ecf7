#!/bin/bash
# NOTE: needs the kernel of profiles/probes/w8_half128_producer_waves_v2.patch (or ..._balanced_k_ranges_v1.patch) applied and built first:
#   git apply profiles/probes/w8_half128_producer_waves_v2.patch && make -C ppl.llm.serving_amd/csrc     (measured, not adopted: profiles/r05_w8_midbatch.md)
# round 5: cycle stamps of every wave of three blocks of gemm_w8_ranges_kernel (-DRG_TIME_BUILD; PPLHIP_RG_TIME=1 prints a warm call's
# stamps relative to each wave's start): where does an iteration's time go?   usage: M=8 bash profiles/probes/w8_ranges_timeline.sh
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
C=ppl.llm.serving_amd/csrc
mkdir -p /tmp/rt gpurun_out
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -mllvm -amdgpu-mfma-vgpr-form=1"
/opt/rocm/bin/hipcc $FLAGS -DRG_TIME_BUILD -c $C/k_gemm_half128.hip -o /tmp/rt/k_gemm_half128.o || exit 1
OBJS=$(cd $C && ls *.o | grep -v k_gemm_half128.o | sed "s#^#$C/#")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/rt/libpplhip.so $OBJS /tmp/rt/k_gemm_half128.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib || exit 1
for m in ${MS:-8 64}; do
  PPLHIP_LIB=/tmp/rt/libpplhip.so PPLHIP_RG_TIME=1 python profiles/gemv_microbench.py 8 $m 2>&1 | grep "rg_time\|^M="
done > gpurun_out/w8_ranges_timeline.log 2>&1
python3 profiles/probes/w8_ranges_timeline_summary.py gpurun_out/w8_ranges_timeline.log
