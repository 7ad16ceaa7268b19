#!/bin/bash
# mixed-block handling on ONE box.  z0 = before this step (indicator MFMAs on every dirty tile, exchange rows may exceed the grid);
# cur = sparse missing-genotype gather (GM_SPARSE_ZMAX 8) + batches cut so that the exchange rows fit the grid
for rep in 1 2; do
for cfg in "z0 c6" "cur c6" "z0 c5" "cur c5" "cur c3" "curmixed c3"; do set -- $cfg
  lib=gmrm_amd/libgmrm_hip.so; [ $1 = z0 ] && lib=gmrm_amd/libgmrm_hip_z0.so
  unset GMRM_FORCE_MIXED; [ $1 = curmixed ] && export GMRM_FORCE_MIXED=1
  v=$(GMRM_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --workload $2 --steps 4 --warmup 4 --no-cpu-baseline --no-signal 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms_avg'],2), [round(x,1) for x in d['roofline']['kernel_ms_per_launch']], d['sweep']['sync_rounds_per_sweep'][-1])")
  echo "$cfg rep$rep: $v"
done; done
