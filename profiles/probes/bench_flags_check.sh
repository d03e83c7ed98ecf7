cd $GRAFT_REPO_ROOT
L="--no-cpu-baseline --no-serving-leg --no-i8i8-leg --prefill-sample 0 --ragged-steps 0"
for f in "--emulate-tp 8 --breakdown-steps 0" "--breakdown" "--cache-mode 1 --breakdown-steps 0" "--model llama2-70b --weight-quant 4 --batch 256 --kv-len 2048 --emulate-tp 8 --breakdown-steps 0" "--act-quant 8 --breakdown-steps 0"; do
  echo "== $f"; python bench.py --steps 4 --warmup 1 $L $f 2>/tmp/err.log | python -c "import sys,json; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(r['ms_per_step'], r['roofline']['frac'], r['breakdown_ms_per_step']['gemm'], r.get('INVALID','')[:40])" || tail -3 /tmp/err.log
done
echo "== default legs (ragged + prefill sample), no serving"; python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-serving-leg --no-i8i8-leg 2>/dev/null | python -c "import sys,json; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(r['ms_per_step'], r['ragged_batch']['ms_per_step'], r['prefill_step_ms'], r['roofline']['traffic'], r['roofline']['traffic_source'][:60])"
