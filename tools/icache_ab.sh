#!/bin/bash
# instruction-cache counters of k_sweep for several library builds on one box: tools/icache_ab.sh libA.so libB.so ...
cd /tmp && export TMPDIR=/tmp
ROOT=/root/repo
for lib in "$@"; do
  rm -rf /tmp/ic_out
  GMRM_HIP_LIB=$ROOT/$lib timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/ic_out -- python3 $ROOT/bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-signal > /dev/null 2> /tmp/ic_err.txt || { tail -3 /tmp/ic_err.txt; continue; }
  python3 - "$lib" <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/ic_out/**/*counter_collection.csv', recursive=True)[0]
agg = {}
for r in csv.DictReader(open(f)):
    if 'k_sweep' not in r['Kernel_Name']: continue
    k = (int(r['Dispatch_Id']), r['Counter_Name'])
    agg[k] = agg.get(k, 0.0) + float(r['Counter_Value'])
last = max(d for d, _ in agg)
print(sys.argv[1], {c: v for (d, c), v in agg.items() if d == last})
PY
done
