#!/bin/bash
# rocprofv3 kernel times of the --predict kernels at full size (tools/bench_predict.py), without and with missing genotypes;
# summaries -> gpurun_out/predict_kernel_stats.csv, gpurun_out/predict_missing_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pp -- python3 $R/tools/bench_predict.py > $R/gpurun_out/predict_bench.json 2> $R/gpurun_out/pp.err || exit 1
cp "$(find $R/gpurun_out/pp -name '*kernel_stats.csv' | head -1)" $R/gpurun_out/predict_kernel_stats.csv
rm -rf $R/gpurun_out/pp
grep -i "k_pg\|predict_g\|assoc\|marker_stats" $R/gpurun_out/predict_kernel_stats.csv
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pp -- python3 $R/tools/bench_predict.py --missing 0.01 > $R/gpurun_out/predict_missing_bench.json 2> $R/gpurun_out/pp.err || exit 1
cp "$(find $R/gpurun_out/pp -name '*kernel_stats.csv' | head -1)" $R/gpurun_out/predict_missing_kernel_stats.csv
rm -rf $R/gpurun_out/pp
grep -i "k_pg\|predict_g\|assoc\|marker_stats" $R/gpurun_out/predict_missing_kernel_stats.csv
