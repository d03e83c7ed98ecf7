cd $GRAFT_REPO_ROOT
L="--no-cpu-baseline --no-serving-leg --no-i8i8-leg --prefill-sample 0 --ragged-steps 0 --breakdown-steps 0"
for tp in 8 4; do for ks in 0 1; do for fn in 0 1; do
  for rep in 1 2; do
  PPLHIP_GEMM_KS=$ks PPLHIP_TP_FUSE_NORM=$fn python bench.py --emulate-tp $tp $L 2>/dev/null | python -c "import sys,json; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('tp$tp ks=$ks fuse=$fn', r['ms_per_step'])"
  done
done; done; done
