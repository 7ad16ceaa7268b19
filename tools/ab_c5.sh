for rep in 1 2; do for lib in gmrm_amd/libgmrm_hip_pre.so gmrm_amd/libgmrm_hip.so; do
  v=$(GMRM_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --workload c5 --steps 4 --warmup 3 --no-cpu-baseline --no-signal 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms_avg'],2), [round(x,1) for x in d['roofline']['kernel_ms_per_launch']], d['sweep']['sync_rounds_per_sweep'][-1])")
  echo "$lib rep$rep: $v"
done; done
