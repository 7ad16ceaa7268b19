#!/bin/bash
# do the sweeps of several phenotypes overlap on the device?  kernel trace of `bench.py --workload $1 --traits $2`
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tt_out
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tt_out -- python3 /root/repo/bench.py --workload ${1:-c2} --traits ${2:-4} --steps 3 --warmup 3 --no-cpu-baseline --no-signal > /tmp/tt_bench.json 2> /tmp/tt_err.txt || { tail -5 /tmp/tt_err.txt; exit 1; }
python3 - <<'PY'
import csv, glob, json
f = glob.glob('/tmp/tt_out/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_sweep' in r['Kernel_Name']]
t0 = int(rows[0]['Start_Timestamp'])
for r in rows[-12:]:
    print(f"{(int(r['Start_Timestamp'])-t0)/1e6:10.3f} ms -> {(int(r['End_Timestamp'])-t0)/1e6:10.3f} ms  ({(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6:7.3f} ms)  stream {r.get('Stream_Id','?')} queue {r.get('Queue_Id','?')}  grid {r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size','?')}")
d = json.loads(open('/tmp/tt_bench.json').read())
print('ms_per_step', round(d['ms_per_step'], 2), 'kernel_ms_avg', round(d['roofline']['kernel_ms_avg'], 2), 'value', round(d['value']))
PY
