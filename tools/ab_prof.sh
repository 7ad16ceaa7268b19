#!/bin/bash
# phase stamps of two diagnostic builds on one box: tools/ab_prof.sh libA_prof.so libB_prof.so [bench args]
A=$1; B=$2; shift 2
for lib in $A $B; do
  echo "== $lib"
  GMRM_HIP_LIB=$PWD/$lib GMRM_SWEEP_PROF=1 timeout -k 10 300 python bench.py --steps 1 --warmup 5 --no-cpu-baseline --no-signal "$@" 2>&1 >/dev/null | grep "sweep prof" | tail -4
done
