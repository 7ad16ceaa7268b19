#!/bin/bash
# Everything profiles/ quotes for one round, in one GPU call:  bash tools/profile_round.sh r02
# (bench lines of every BASELINE configuration that fits one GPU, rocprofv3 kernel stats of the default
#  bench command, separate --pmc passes for FETCH_SIZE / WRITE_SIZE condensed by tools/pmc_summary.py)
set -o pipefail
TAG=${1:-r02}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py"
echo "== bench c3 (default: 2 warm-up + 5 timed sweeps, CPU baseline, signal-bearing block)"
timeout -k 10 400 $B > $OUT/bench_c3.json 2> $OUT/bench_c3.err || exit 1
echo "== bench c3 as the driver runs it (--steps 20 --warmup 5)"
timeout -k 10 400 $B --steps 20 --warmup 5 --no-cpu-baseline --no-signal > $OUT/bench_c3_20steps.json 2>> $OUT/bench_c3.err || exit 1
for wl in c2 c4 c5 c6; do
  echo "== bench $wl"
  timeout -k 10 400 $B --workload $wl --no-cpu-baseline --no-signal > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err || exit 1
done
echo "== rocprofv3 --kernel-trace --stats"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python3 $ROOT/bench.py --no-cpu-baseline --no-signal > $OUT/bench_under_rocprof.json 2> $OUT/rocprof_stats.err || exit 1
cp "$(find $OUT/prof_stats -name '*kernel_stats.csv' | head -1)" $OUT/${TAG}_kernel_stats.csv
echo "== rocprofv3 --pmc FETCH_SIZE"
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 2 --warmup 5 --no-cpu-baseline --no-signal > $OUT/bench_under_pmc_fetch.json 2> $OUT/rocprof_fetch.err || exit 1
echo "== rocprofv3 --pmc WRITE_SIZE"
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --steps 2 --warmup 5 --no-cpu-baseline --no-signal > $OUT/bench_under_pmc_write.json 2> $OUT/rocprof_write.err || exit 1
python3 $ROOT/tools/pmc_summary.py "$(find $OUT/pmc_fetch -name '*counter_collection.csv' | head -1)" "$(find $OUT/pmc_write -name '*counter_collection.csv' | head -1)" \
  $OUT/$TAG 2 "bench.py --steps 2 --warmup 5 (c3: 500 000 x 1 000 000; the two timed launches are stationary sweeps 6 and 7), separate rocprofv3 --pmc passes" || exit 1
rm -rf $OUT/prof_stats $OUT/pmc_fetch $OUT/pmc_write
ls -la $OUT
echo "== phase stamps (diagnostic build)"
GMRM_HIP_LIB=$ROOT/gmrm_amd/libgmrm_hip_prof.so GMRM_SWEEP_PROF=1 timeout -k 10 300 python3 $ROOT/bench.py --steps 1 --warmup 5 --no-cpu-baseline --no-signal 2>&1 >/dev/null | grep "sweep prof" > $OUT/${TAG}_phase_stamps.txt
cat $OUT/${TAG}_phase_stamps.txt
