#!/bin/bash
# FETCH_SIZE of a stationary k_sweep launch and its time against how far ahead the genotype prefetch requests
# (GMRM_PF_AHEAD16: sixteenths of the current batch assumed walked; what does not fit the ring window is dropped and requested again)
cd /tmp && export TMPDIR=/tmp
for ah in 16 12 8 4 0; do
  rm -rf /tmp/tr_out
  GMRM_PF_AHEAD16=$ah timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/tr_out -- python3 /root/repo/bench.py --steps 2 --warmup 5 --no-cpu-baseline --no-signal > /tmp/tr_bench.json 2> /tmp/tr_err.txt || { tail -3 /tmp/tr_err.txt; continue; }
  GMRM_PF_AHEAD16=$ah timeout -k 10 300 python3 /root/repo/bench.py --steps 6 --warmup 5 --no-cpu-baseline --no-signal > /tmp/tr_bench2.json 2>/dev/null
  python3 - $ah <<'PY'
import csv, glob, sys, json
f = glob.glob('/tmp/tr_out/**/*counter_collection.csv', recursive=True)[0]
agg = {}
for r in csv.DictReader(open(f)):
    if 'k_sweep' not in r['Kernel_Name']: continue
    k = int(r['Dispatch_Id'])
    agg[k] = agg.get(k, 0.0) + float(r['Counter_Value'])
last = sorted(agg)[-2:]
d = json.loads(open('/tmp/tr_bench2.json').read())
gb = [2 * agg[k] * 1024 / 1e9 for k in last]          # x2: the gfx950 correction for 16-byte-per-lane loads (guide, HBM section)
print('ahead16', sys.argv[1], 'fetch GB per launch (corrected)', [round(x, 1) for x in gb], 'x algorithmic', round(sum(gb) / len(gb) / 125.0, 3), 'kernel_ms (no profiler)', round(d['roofline']['kernel_ms_avg'], 2))
PY
done
