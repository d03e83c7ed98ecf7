for x in 0 1 2; do for d in 0 2; do for tp in 2 8; do
  echo "== xalloc=$x debug=$d tp=$tp"
  PPLHIP_P2P_XALLOC=$x PPLHIP_TP_DEBUG=$d timeout 120 python profiles/probes/tp_overlap_debug.py $tp 0 2>&1 | grep -v amdgpu.ids | awk '{ if ($6+0 > 0.008) bad++; n++ } END { print "steps", n, "bad", bad+0 }'
done; done; done
