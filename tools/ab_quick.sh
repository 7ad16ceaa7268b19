#!/bin/bash
# A/B on one box: the committed kernel (libgmrm_hip_pre.so, tools/build_variant.py pre HEAD), the working tree's, the working
# tree's with the continuation switched off; then the working tree's phase stamps (diagnostic build)
run() { timeout -k 10 200 python bench.py --steps 8 --warmup 5 --no-cpu-baseline --no-signal $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms_avg'],2), 'rounds', d['sweep']['sync_rounds_per_sweep'][-1], 'crossed', d['sweep'].get('crossed_stops_per_sweep',[0])[-1], 'warm', [round(x,1) for x in d['roofline']['kernel_ms_warmup_launches'][:2]])"; }
for rep in 1 2; do
  echo "pre     rep$rep: $(GMRM_HIP_LIB=$PWD/gmrm_amd/libgmrm_hip_pre.so run)"
  echo "new     rep$rep: $(run)"
  echo "nocross rep$rep: $(GMRM_NO_CROSS=1 run)"
done
for nc in 0 1; do
  echo "== phase stamps, GMRM_NO_CROSS=$nc"
  if [ $nc = 1 ]; then export GMRM_NO_CROSS=1; fi
  GMRM_HIP_LIB=$PWD/gmrm_amd/libgmrm_hip_prof.so GMRM_SWEEP_PROF=1 timeout -k 10 300 python bench.py --steps 1 --warmup 5 --no-cpu-baseline --no-signal $EXTRA 2>&1 >/dev/null | grep "sweep prof" | tail -4
done
