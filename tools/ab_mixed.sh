#!/bin/bash
# the per-marker-layout kernel against its predecessors on ONE box: c5 (every marker dirty), c3 forced through the mixed kernel, c6
for rep in 1 2; do
for cfg in "pre c5" "cur c5" "cur c3" "curmixed c3" "cur c6"; do set -- $cfg
  lib=gmrm_amd/libgmrm_hip.so; [ $1 = pre ] && lib=gmrm_amd/libgmrm_hip_pre.so
  unset GMRM_FORCE_MIXED; [ $1 = curmixed ] && export GMRM_FORCE_MIXED=1
  v=$(GMRM_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --workload $2 --steps 4 --warmup 3 --no-cpu-baseline --no-signal 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms_avg'],2), [round(x,1) for x in d['roofline']['kernel_ms_per_launch']], d['sweep']['sync_rounds_per_sweep'][-1])")
  echo "$cfg rep$rep: $v"
done; done
