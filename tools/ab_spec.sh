#!/bin/bash
# the speculation threshold (GMRM_SPEC_FACTOR16: speculate when the recent run length >= factor/16 batches) on one box:
# tools/ab_spec.sh <lib.so> <workload> <factor> [<factor> ...]
LIB=$1; WL=$2; shift 2
for rep in 1 2; do for f in "$@"; do
  v=$(GMRM_SPEC_FACTOR16=$f GMRM_HIP_LIB=$PWD/$LIB timeout -k 10 200 python bench.py --workload $WL --steps 6 --warmup 5 --no-cpu-baseline --no-signal 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms_avg'],2), 'rounds', d['sweep']['sync_rounds_per_sweep'][-1], 'discarded', d['sweep'].get('discarded_batches_per_sweep'))")
  echo "$LIB $WL factor $f rep$rep: $v"
done; done
