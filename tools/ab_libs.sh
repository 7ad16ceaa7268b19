#!/bin/bash
# kernel time of several library builds on one box, two repetitions: tools/ab_libs.sh [bench args --] libA.so libB.so ...
ARGS=""
while [ "$1" != "--" ] && [[ "$1" != *.so ]]; do ARGS="$ARGS $1"; shift; done
[ "$1" = "--" ] && shift
for rep in 1 2; do for lib in "$@"; do
  v=$(GMRM_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 6 --warmup 5 --no-cpu-baseline --no-signal $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms_avg'],2), 'rounds', d['sweep']['sync_rounds_per_sweep'][-1], 'warm', [round(x,1) for x in d['roofline']['kernel_ms_warmup_launches'][:3]])")
  echo "$lib rep$rep: $v"
done; done
