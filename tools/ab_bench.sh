#!/bin/bash
# A/B of library builds on ONE box (devices differ by several per cent): tools/ab_bench.sh libA.so libB.so ...
for rep in 1 2; do for lib in "$@"; do
  v=$(GMRM_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 5 --warmup 4 --no-cpu-baseline --no-signal 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms_avg'],2), [round(x,1) for x in d['roofline']['kernel_ms_per_launch']])")
  echo "$lib rep$rep: $v"
done; done
