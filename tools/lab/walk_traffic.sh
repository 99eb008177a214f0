#!/bin/bash
# HBM fetch per SpMV of the unsplit wave kernel under the static chunked walk and the ordered per-XCD walk (QBH_WAVE_SWIZZLE)
set -u
R=/root/repo
export TMPDIR=/tmp QBH_NO_AUTOTUNE=1 SPMV_REPS=4
W=${1:-triangular_6x6_k10_n15}
cd /tmp
for m in 2 3; do
  export QBH_WAVE_SWIZZLE=$m
  rm -rf /tmp/wt$m; mkdir -p /tmp/wt$m
  timeout 900 rocprofv3 --pmc FETCH_SIZE -d /tmp/wt$m/g1 -o p -- python3 $R/tools/spmv_time.py $W "" > /tmp/wt$m/log 2>&1
  grep "ms/launch" /tmp/wt$m/log
  python3 $R/tools/pmc_summary.py /tmp/wt$m "%k_spmv_%"
done
