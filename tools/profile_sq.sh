#!/bin/bash
# SQ counter evidence for the sweep kernel's bound:  bash tools/profile_sq.sh r03 [workloads...]
# One rocprofv3 --pmc pass per group of <= 8 SQ counters (the SQ block has 8 slots per pass on gfx950), the
# program directly after `--`, --kernel-trace/--stats only.  tools/pmc_sq_summary.py condenses the passes.
set -o pipefail
TAG=${1:-r03}
shift
WLS=${@:-c3 c5}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PASS_A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
PASS_B="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"
PASS_C="SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES"
for wl in $WLS; do
  for p in A B C; do
    eval "CNT=\$PASS_$p"
    echo "== $wl pass $p: $CNT"
    timeout -k 10 400 rocprofv3 --pmc $CNT --output-format csv -d $OUT/sq_${wl}_$p -- python3 $ROOT/bench.py --workload $wl --steps 2 --warmup 5 --no-cpu-baseline --no-signal \
      > $OUT/bench_under_sq_${wl}_$p.json 2> $OUT/rocprof_sq_${wl}_$p.err || { tail -5 $OUT/rocprof_sq_${wl}_$p.err; exit 1; }
  done
  python3 $ROOT/tools/pmc_sq_summary.py $OUT/${TAG}_pmc_sq_${wl}.json 2 $wl \
    $(for p in A B C; do find $OUT/sq_${wl}_$p -name '*counter_collection.csv' | head -1; done) || exit 1
  rm -rf $OUT/sq_${wl}_A $OUT/sq_${wl}_B $OUT/sq_${wl}_C
done
ls -la $OUT
