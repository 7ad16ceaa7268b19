#!/bin/bash
# Diagnostic builds whose in-kernel stamps are taken by wavefront 0, 1, 2 and 3 (thread 0, 64, 128, 192):
# gmrm_amd/libgmrm_hip_prof_w{0,1,2,3}.so.  tools/build_prof_waves.sh [waves...]
set -e
cd "$(dirname "$0")/.."
WAVES=${@:-0 1 2 3}
python -m gmrm_amd.build --prof >/dev/null          # all objects (stamps by thread 0) in gmrm_amd/_build_prof
D=gmrm_amd/_build_prof
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wall -Wno-unused-function -DGM_SWEEP_PROF"
pids=()
for w in $WAVES; do
  if [ "$w" = 0 ]; then cp gmrm_amd/libgmrm_hip_prof.so gmrm_amd/libgmrm_hip_prof_w0.so; continue; fi
  hipcc $FLAGS -DGM_PROF_TID=$((64 * w)) -c gmrm_amd/csrc/sweep.hip -o $D/sweep_w$w.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
for w in $WAVES; do
  [ "$w" = 0 ] && continue
  hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o gmrm_amd/libgmrm_hip_prof_w$w.so $D/ops_hip.o $D/sweep_w$w.o $D/capi_cpp.o $D/sampler_cpp.o $D/ingest_cpp.o $D/shard_group_cpp.o
done
ls -la gmrm_amd/libgmrm_hip_prof_w*.so
