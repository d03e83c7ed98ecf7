#!/bin/bash
# builds profiles/probes/libgemm_asm_probe[_ablN].so from profiles/probes/gemm_asm/k_gemm_asm.hip (N: generator ablation bits 1 | 2 | 4, kernel bits 4 | 8)
# (round 6: the hand-scheduled kernel, its generated K loop and the generator moved here from csrc/ -- equal speed to the product kernel, no default
#  path reached it; `python3 gen_gemm_asm.py 3 > k_gemm_asm_ni3.inc` regenerates the committed loop)
set -e
cd "$(dirname "$0")/gemm_asm"
OUT=..
FLAGS="-I../../../ppl.llm.serving_amd/csrc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-value -shared -DGA_PROBE_BUILD"
/opt/rocm/bin/hipcc $FLAGS k_gemm_asm.hip -o $OUT/libgemm_asm_probe.so
# variants: <abl>[:<stages>[:<prio>[:direct]]]   (direct = the unstaged epilogue)
for v in "$@"; do
  IFS=: read a st pr ex <<< "$v"; st=${st:-3}; pr=${pr:-0}
  tag=abl${a}_s${st}_p${pr}${ex:+_$ex}
  EXTRA=""; [ "$ex" = "direct" ] && EXTRA="-DGA_NO_STAGE"
  python3 gen_gemm_asm.py 3 $((a & 7)) $st $pr > /tmp/ga_$tag.inc
  /opt/rocm/bin/hipcc $FLAGS -DGA_INC="\"/tmp/ga_$tag.inc\"" -DGA_ABL=$a -DGA_STAGES=$st $EXTRA k_gemm_asm.hip -o $OUT/libgemm_asm_probe_$tag.so
done
