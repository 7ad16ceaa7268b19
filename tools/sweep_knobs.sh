for nbf in 16 24 32 48 64; do for sp in 12 24 48; do
  v=$(GMRM_NB_FACTOR16=$nbf GMRM_SPEC_FACTOR16=$sp timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['kernel_ms_avg'],1), d['sweep']['sync_rounds_per_sweep'])")
  echo "nbf=$nbf spec=$sp -> $v"
done; done
