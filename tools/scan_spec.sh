GMRM_HIP_LIB=$PWD/gmrm_amd/libgmrm_hip_prof.so GMRM_SWEEP_PROF=1 timeout -k 10 300 python bench.py --steps 2 --warmup 4 --no-cpu-baseline --no-signal > gpurun_out/bench_prof2.json 2> gpurun_out/bench_prof2.err
tail -4 gpurun_out/bench_prof2.err
for sp in 4 8 16 32; do for nbf in 24 32; do
  v=$(GMRM_NB_FACTOR16=$nbf GMRM_SPEC_FACTOR16=$sp timeout -k 10 200 python bench.py --steps 4 --warmup 4 --no-cpu-baseline --no-signal 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['kernel_ms_avg'],1), d['sweep']['sync_rounds_per_sweep'])")
  echo "spec=$sp nbf=$nbf -> $v"
done; done
