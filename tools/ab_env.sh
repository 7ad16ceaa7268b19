#!/bin/bash
# A/B of environment knobs on ONE box (devices differ by several per cent):  tools/ab_env.sh "" "GMRM_NO_GATE=1" ...   [env BENCH_ARGS="--workload c5"]
for rep in 1 2; do for e in "$@"; do
  v=$(env $e timeout -k 10 200 python bench.py --steps 6 --warmup 5 --no-cpu-baseline --no-signal $BENCH_ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['roofline']['kernel_ms_avg'],2), d['sweep']['sync_rounds_per_sweep'][-1], [round(x,1) for x in d['roofline']['kernel_ms_per_launch']])")
  echo "[$e] rep$rep: $v"
done; done
