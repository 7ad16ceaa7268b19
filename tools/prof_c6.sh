#!/bin/bash
# in-kernel phase stamps (diagnostic build, python -m gmrm_amd.build --prof) of c6, c3 forced through the mixed-layout kernel and c3,
# then the plain timings of the three
export GMRM_HIP_LIB=$PWD/gmrm_amd/libgmrm_hip_prof.so GMRM_SWEEP_PROF=1
for cfg in "c6 0" "c3 1" "c3 0"; do set -- $cfg
  unset GMRM_FORCE_MIXED; [ $2 = 1 ] && export GMRM_FORCE_MIXED=1
  echo "== $1 forced_mixed=$2"
  timeout -k 10 300 python bench.py --workload $1 --steps 1 --warmup 4 --no-cpu-baseline --no-signal 2>&1 | grep -v "^{" | tail -4
done
unset GMRM_HIP_LIB GMRM_SWEEP_PROF GMRM_FORCE_MIXED
for cfg in "c6 0" "c3 1" "c3 0"; do set -- $cfg
  unset GMRM_FORCE_MIXED; [ $2 = 1 ] && export GMRM_FORCE_MIXED=1
  timeout -k 10 300 python bench.py --workload $1 --steps 4 --warmup 4 --no-cpu-baseline --no-signal 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 forced_mixed=$2', round(d['roofline']['kernel_ms_avg'],2), [round(x,1) for x in d['roofline']['kernel_ms_per_launch']], d['sweep']['sync_rounds_per_sweep'][-1])"
done
