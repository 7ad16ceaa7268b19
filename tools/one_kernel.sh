#!/bin/bash
# Code-generation experiments on the headline kernel alone: compiles k_sweep<2,0,false> only (-DGM_ONE_KERNEL) to assembly
# and prints its resource numbers and scratch traffic.   tools/one_kernel.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wno-unused-function -DGM_ONE_KERNEL "$@" -S --cuda-device-only gmrm_amd/csrc/sweep.hip -o /tmp/one_kernel.s 2>&1 | grep -E "error|warning: [^a]" || true
awk "/^_ZN2gm7k_sweepILi2ELi[01]ELb[01]/,/\\.Lfunc_end/" /tmp/one_kernel.s > /tmp/one_kernel_body.s
echo "lines $(wc -l < /tmp/one_kernel_body.s)  scratch ops $(grep -c scratch_ /tmp/one_kernel_body.s || true)  mfma $(grep -c v_mfma /tmp/one_kernel_body.s)  glds $(grep -c global_load_lds /tmp/one_kernel_body.s)"
grep -E "^\s+\.(vgpr_count|agpr_count|sgpr_spill_count|vgpr_spill_count|private_segment_fixed_size):" /tmp/one_kernel.s | head -5 | tr -s ' ' | tr '\n' ' '; echo
