#!/bin/bash
# What one GPU's share of the 500k x 1M problem costs when it is 1/1, 1/2, 1/4, 1/8 of the markers (VERDICT r2 next #2a),
# the c2 geometry with 1 and 4 phenotypes side by side (next #4), and a one-rank RCCL all-reduce of the 8 MB exchange buffer.
#   bash tools/shard_sized.sh r03
set -o pipefail
TAG=${1:-r03}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for m in 1000000 500000 250000 125000; do
  echo "== c3 geometry, $m markers on this GPU"
  timeout -k 10 300 python3 $ROOT/bench.py --markers $m --steps 10 --warmup 5 --no-cpu-baseline --no-signal > $OUT/shard_$m.json 2> $OUT/shard_$m.err || { tail -5 $OUT/shard_$m.err; exit 1; }
done
echo "== the sharded driver's own cost: one rank under torchrun (RCCL), 125000 markers, residual exchange forced (an identity with one rank)"
GMRM_BENCH_FORCE_EXCHANGE=1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 \
  $ROOT/bench.py --gpus 1 --markers 125000 --steps 20 --warmup 5 --no-cpu-baseline --no-signal > $OUT/driver_125000.json 2> $OUT/driver_125000.err || { tail -5 $OUT/driver_125000.err; exit 1; }
for t in 1 4; do
  echo "== c2 geometry, $t phenotype(s)"
  timeout -k 10 300 python3 $ROOT/bench.py --workload c2 --traits $t --steps 10 --warmup 5 --no-cpu-baseline --no-signal > $OUT/c2_T$t.json 2> $OUT/c2_T$t.err || { tail -5 $OUT/c2_T$t.err; exit 1; }
done
echo "== one-rank RCCL all-reduce of the residual exchange buffer (2 x 4 MB f64)"
MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 200 python3 - > $OUT/rccl_one_rank.json 2> $OUT/rccl_one_rank.err <<'PY' || { tail -5 $OUT/rccl_one_rank.err; exit 1; }
import json, time, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
x = torch.zeros(2 * 500_000, dtype=torch.float64, device="cuda")
for _ in range(5):
    dist.all_reduce(x)
torch.cuda.synchronize()
ts = []
for _ in range(50):
    t0 = time.perf_counter(); dist.all_reduce(x); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
ts.sort()
print(json.dumps({"what": "torch.distributed all_reduce(SUM) of 1e6 f64 (8 MB), backend nccl = RCCL, ONE rank, host wall incl. synchronize",
                  "median_us": ts[len(ts) // 2] * 1e6, "min_us": ts[0] * 1e6, "max_us": ts[-1] * 1e6}))
dist.destroy_process_group()
PY
python3 - $OUT <<'PY'
import json, sys, glob, os
out = sys.argv[1]
rows = []
for f in sorted(glob.glob(out + "/shard_[0-9]*.json"), key=lambda p: -int(os.path.basename(p)[6:-5])):
    d = json.load(open(f))
    rows.append({"markers": d["config"]["markers_per_gpu"], "ms_per_step": d["ms_per_step"], "kernel_ms_avg": d["roofline"]["kernel_ms_avg"],
                 "host_ms_per_step": d["ms_per_step"] - d["roofline"]["kernel_ms_avg"], "rounds": d["sweep"]["sync_rounds_per_sweep"],
                 "updates": d["sweep"]["updates_per_sweep"], "value": d["value"], "frac": d["roofline"]["frac"]})
c2 = {}
for t in (1, 4):
    d = json.load(open(out + "/c2_T%d.json" % t))
    c2["T%d" % t] = {"value": d["value"], "ms_per_step": d["ms_per_step"], "kernel_ms_avg": d["roofline"]["kernel_ms_avg"]}
rc = json.loads([l for l in open(out + "/rccl_one_rank.json") if l.startswith("{")][-1])   # RCCL prints a banner first
dd = json.loads([l for l in open(out + "/driver_125000.json") if l.startswith("{")][-1])
drv = {"what": "bench.py --gpus 1 --markers 125000 under torchrun: ShardedDriver with RCCL, one rank, residual exchange forced "
               "(delta export -> all-reduce of the packed buffer -> import; mu and hyper broadcasts)",
       "ms_per_step": dd["ms_per_step"], "kernel_ms_avg": dd["roofline"]["kernel_ms_avg"],
       "host_and_exchange_ms_per_step": dd["ms_per_step"] - dd["roofline"]["kernel_ms_avg"],
       "collectives_per_sweep": dd.get("collectives_per_sweep")}
json.dump({"shard_sized": rows, "c2_traits": c2, "rccl_one_rank": rc, "sharded_driver_one_rank": drv}, open(out + "/shard_sized.json", "w"), indent=1)
print(json.dumps(drv, indent=1))
print(json.dumps({"shard_sized": [{k: (v if not isinstance(v, list) else v[-1]) for k, v in r.items()} for r in rows], "c2": c2, "rccl": rc}, indent=1))
PY
