# stationary-regime knob scan: 4 warm-up sweeps, 3 timed
for sp in 8 16 24 32 48 64; do
  v=$(GMRM_SPEC_FACTOR16=$sp timeout -k 10 200 python bench.py --steps 3 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['kernel_ms_avg'],1), d['sweep']['sync_rounds_per_sweep'], [round(x) for x in d['roofline']['kernel_ms_warmup_launches']])")
  echo "spec=$sp -> $v"
done
