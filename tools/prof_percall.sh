#!/bin/bash
# rocprofv3 kernel times of the per-call kernels at full width (tools/bench_percall.py); summary -> gpurun_out/percall_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
python3 $R/tools/bench_percall.py > $R/gpurun_out/percall_bench.json || exit 1
cat $R/gpurun_out/percall_bench.json
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pc -- python3 $R/tools/bench_percall.py > /dev/null 2> $R/gpurun_out/pc.err || exit 1
cp "$(find $R/gpurun_out/pc -name '*kernel_stats.csv' | head -1)" $R/gpurun_out/percall_kernel_stats.csv
rm -rf $R/gpurun_out/pc
grep -i "k_dot\|k_update\|k_offset\|k_sumsq" $R/gpurun_out/percall_kernel_stats.csv
