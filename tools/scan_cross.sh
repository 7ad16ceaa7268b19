#!/bin/bash
# kernel time against the smallest number of followers for which the walk crosses a stop (GMRM_CROSS_FRAC16, sixteenths of the batch), one box
run() { timeout -k 10 200 python bench.py --steps 6 --warmup 5 --no-cpu-baseline --no-signal "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms_avg'],2), 'rounds', d['sweep']['sync_rounds_per_sweep'][-1], 'crossed', d['sweep']['crossed_stops_per_sweep'][-1], 'warm', [round(x,1) for x in d['roofline']['kernel_ms_warmup_launches'][:3]])"; }
echo "nocross: $(GMRM_NO_CROSS=1 run "$@")"
for t in 1 4 6 8 9 10 12; do echo "cross_frac16=$t: $(GMRM_CROSS_FRAC16=$t run "$@")"; done
echo "nocross: $(GMRM_NO_CROSS=1 run "$@")"
