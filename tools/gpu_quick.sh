#!/bin/bash
# quick GPU check used while iterating on the sweep kernel: chain parity tests, a short bench, phase stamps
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_configs.py::test_c4_four_traits_at_full_width_run_as_queued_pairs -m gpu -x -q 2>&1 | tail -3 || exit 1
timeout -k 10 300 python bench.py --steps 5 --warmup 4 --no-cpu-baseline --no-signal 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value', round(d['value']), 'kernel_ms', [round(x,1) for x in d['roofline']['kernel_ms_per_launch']], 'warm', [round(x,1) for x in d['roofline']['kernel_ms_warmup_launches']], 'rounds', d['sweep']['sync_rounds_per_sweep'][-1])"
GMRM_HIP_LIB=$PWD/gmrm_amd/libgmrm_hip_prof.so GMRM_SWEEP_PROF=1 timeout -k 10 300 python bench.py --steps 1 --warmup 5 --no-cpu-baseline --no-signal 2>&1 >/dev/null | grep "sweep prof" | tail -4
