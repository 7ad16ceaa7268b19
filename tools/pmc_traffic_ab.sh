#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of a stationary k_sweep launch for two column-stride alignments on one box
cd /tmp && export TMPDIR=/tmp
for al in 16 128; do
  for cn in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/tr_out
    GMRM_STRIDE_ALIGN=$al timeout -k 10 300 rocprofv3 --pmc $cn --output-format csv -d /tmp/tr_out -- python3 /root/repo/bench.py --steps 2 --warmup 5 --no-cpu-baseline --no-signal > /tmp/tr_bench.json 2> /tmp/tr_err.txt || { tail -3 /tmp/tr_err.txt; continue; }
    python3 - $al $cn <<'PY'
import csv, glob, sys, json
f = glob.glob('/tmp/tr_out/**/*counter_collection.csv', recursive=True)[0]
agg = {}
for r in csv.DictReader(open(f)):
    if 'k_sweep' not in r['Kernel_Name']: continue
    k = int(r['Dispatch_Id'])
    agg[k] = agg.get(k, 0.0) + float(r['Counter_Value'])
last = sorted(agg)[-2:]
d = json.loads(open('/tmp/tr_bench.json').read())
print('align', sys.argv[1], sys.argv[2], 'KiB per launch', [agg[k] for k in last], 'GB', [round(agg[k]*1024/1e9,1) for k in last], 'kernel_ms', round(d['roofline']['kernel_ms_avg'],2))
PY
  done
done
