# NOTE: needs profiles/probes/w4_interleaved_nibbles_probe.patch applied (the -DPC_INTERLEAVED_NIBBLES conversion)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
C=ppl.llm.serving_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden"
OBJS=$(cd $C && ls *.o | grep -v k_gemm_pc.o | sed "s#^#$C/#")
for v in natural interleaved; do
  mkdir -p /tmp/pt_$v
  X=""; [ $v = interleaved ] && X="-DPC_INTERLEAVED_NIBBLES"
  /opt/rocm/bin/hipcc $FLAGS -DPC_TIME_BUILD $X -c $C/k_gemm_pc.hip -o /tmp/pt_$v/k_gemm_pc.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/pt_$v/libpplhip.so $OBJS /tmp/pt_$v/k_gemm_pc.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib || exit 1
  echo "== $v"
  PPLHIP_LIB=/tmp/pt_$v/libpplhip.so PPLHIP_PC_TIME=1 python profiles/gemm_microbench.py 256 4 70b-tp8 w13 2>&1 | grep "pc_time.*wave 0 \|pc_time.*wave 4 \|^w13"
done
