#!/bin/bash
# In-kernel stamps of the last (stationary) sweep as seen by each wavefront: tools/prof_waves.sh OUT workload [workload ...]
OUT=$1; shift
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
for wl in "$@"; do
  for w in 0 1 2 3; do
    echo "== $wl wavefront $w" >> "$OUT"
    GMRM_HIP_LIB=$PWD/gmrm_amd/libgmrm_hip_prof_w$w.so GMRM_SWEEP_PROF=1 timeout -k 10 300 python bench.py --workload $wl --steps 1 --warmup 5 --no-cpu-baseline --no-signal 2>&1 >/dev/null | grep "sweep prof" | tail -5 >> "$OUT" || exit 1
  done
done
